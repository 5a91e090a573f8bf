// fft_queue_sim.cpp — host simulation of the overlap-save kernel's block queue (TEST INFRASTRUCTURE).
// Compiles qo-100-tools_amd/csrc/if_fir_fft_queue.h — the code the kernel runs — against std::atomic and runs the waves
// of a launch as threads with random delays: every block must be handed out exactly once, every wave must leave, no
// bounded wait may expire, and the ticket counter must stay below the launcher's bound.
// usage: fft_queue_sim <nblocks> <workgroups> <seed> [delay_slot0_taker]
//   delay_slot0_taker = 1: the wave that owes a ticket fetch is held back before it draws (the interleaving ADVICE r2
//   describes: the fetch for local group g+1 overtaken by the one for g+2)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "if_fir_fft_queue.h"

using namespace if_fir;

struct Workgroup
{
    std::atomic<unsigned long long> cur{0};
    std::atomic<unsigned long long> ring[Q_RING];
    std::atomic<unsigned long long> tail{0};
    std::atomic<unsigned> claim[4];
};

static std::atomic<unsigned> g_ticket{0}, g_faults{0}, g_tail_ticket{0};
static unsigned g_wgs = 1;
static bool g_delay_fetch = false;

struct HostQueue
{
    Workgroup *wg;
    std::mt19937 *rng;
    unsigned long long cur_add() { return wg->cur.fetch_add(1, std::memory_order_relaxed); }
    unsigned long long cur_load() { return wg->cur.load(std::memory_order_relaxed); }
    void cur_store(unsigned long long v) { wg->cur.store(v, std::memory_order_relaxed); }
    unsigned long long ring_load(unsigned i) { return wg->ring[i].load(std::memory_order_relaxed); }
    void ring_store(unsigned i, unsigned long long v) { wg->ring[i].store(v, std::memory_order_relaxed); }
    unsigned ticket()
    {
        if (g_delay_fetch && ((*rng)() & 3) == 0)
            std::this_thread::sleep_for(std::chrono::microseconds(200 + (*rng)() % 400));
        return g_ticket.fetch_add(1, std::memory_order_relaxed);
    }
    unsigned tail_claim(unsigned simd) { return wg->claim[simd].fetch_add(1, std::memory_order_relaxed); }
    unsigned long long tail_add() { return wg->tail.fetch_add(1, std::memory_order_relaxed); }
    unsigned long long tail_load() { return wg->tail.load(std::memory_order_relaxed); }
    void tail_store(unsigned long long v) { wg->tail.store(v, std::memory_order_relaxed); }
    unsigned tail_ticket()
    {
        if (g_delay_fetch && ((*rng)() & 1) == 0)
            std::this_thread::sleep_for(std::chrono::microseconds(100 + (*rng)() % 300));
        return g_tail_ticket.fetch_add(1, std::memory_order_relaxed);
    }
    void fault() { g_faults.fetch_add(1); }
    void pause() { std::this_thread::yield(); }
    unsigned wgs() const { return g_wgs; }
};

int main(int argc, char **argv)
{
    if (argc < 4)
    {
        fprintf(stderr, "usage: fft_queue_sim <nblocks> <workgroups> <seed> [delay]\n");
        return 2;
    }
    const int64_t nblocks = atoll(argv[1]);
    const int64_t wgs_max = atoll(argv[2]);
    const unsigned seed = (unsigned)atoi(argv[3]);
    g_delay_fetch = argc > 4 && atoi(argv[4]) != 0;
    // the launcher's grid: one workgroup per CU, never more than there are groups (fft_schedule)
    const int64_t groups = (nblocks + QB - 1) / QB;
    g_wgs = (unsigned)(groups < wgs_max ? groups : wgs_max);
    if (g_wgs < 1)
        g_wgs = 1;
    std::vector<Workgroup> wg(g_wgs);
    for (unsigned b = 0; b < g_wgs; b++)
    {
        wg[b].cur.store(queue_cur_init(b, g_wgs, true)); // local group 0 is taken statically
        for (unsigned i = 0; i < Q_RING; i++)
            wg[b].ring[i].store(queue_ring_init(i, b, g_wgs));
        for (auto &c : wg[b].claim)
            c.store(0);
    }
    const int64_t nmain = queue_main_blocks(nblocks, g_wgs);
    std::vector<std::atomic<int>> tail_per_simd(4 * g_wgs);
    for (auto &t : tail_per_simd)
        t.store(0);
    std::vector<std::atomic<int>> taken(nblocks > 0 ? nblocks : 1);
    for (auto &t : taken)
        t.store(0);
    std::atomic<int> beyond{0};
    std::vector<std::thread> waves;
    for (unsigned b = 0; b < g_wgs; b++)
        for (unsigned w = 0; w < QB; w++)
            waves.emplace_back([&, b, w]() {
                std::mt19937 rng(seed * 7919u + b * 131u + w);
                HostQueue q{&wg[b], &rng};
                // as in the kernel: static first block, wave 0 owes the fetch for local group Q_AHEAD
                int64_t blk = (int64_t)b * QB + w;
                if (w == 0)
                    queue_start(q);
                while (blk < nblocks)
                {
                    if (taken[blk].fetch_add(1) != 0)
                        beyond.fetch_add(1000000);
                    // "process the block": a random amount of time, now and then a long stall
                    const unsigned r = rng();
                    if ((r & 15) == 0)
                        std::this_thread::sleep_for(std::chrono::microseconds(50 + r % 300));
                    else if ((r & 1023) == 1)
                        std::this_thread::sleep_for(std::chrono::milliseconds(3)); // a wave that falls far behind its workgroup
                    else if ((r & 3) == 0)
                        std::this_thread::yield();
                    if (blk >= nmain && tail_per_simd[4 * b + (w & 3)].fetch_add(1) != 0)
                        beyond.fetch_add(1);                 // two tail blocks on one SIMD of one workgroup: never
                    blk = queue_take(q, w & 3, nmain, nblocks);
                }
            });
    for (auto &t : waves)
        t.join();
    int64_t missing = 0, twice = 0;
    for (int64_t i = 0; i < nblocks; i++)
    {
        const int c = taken[i].load();
        missing += c == 0;
        twice += c > 1;
    }
    const unsigned tickets = g_ticket.load();
    const int64_t bound = groups + 2 * (int64_t)g_wgs;
    const bool ok = missing == 0 && twice == 0 && g_faults.load() == 0 && (int64_t)tickets <= bound && beyond.load() == 0 &&
                    (int64_t)g_tail_ticket.load() <= (int64_t)g_wgs;
    printf("nblocks %lld (groups %lld + tail %lld) wgs %u seed %u delay %d: missing %lld twice %lld faults %u tickets %u (bound %lld) "
           "tail tickets %u %s\n", (long long)nblocks, (long long)nmain, (long long)(nblocks - nmain), g_wgs, seed, (int)g_delay_fetch,
           (long long)missing, (long long)twice, g_faults.load(), tickets, (long long)bound, g_tail_ticket.load(), ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
