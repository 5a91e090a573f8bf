// if_fir_fft_queue.h — the two-level block queue of the overlap-save kernel, written once for the device and for a host
// simulation (tests/c/fft_queue_sim.cpp runs the very same code with threads as waves and checks that every block is
// handed out exactly once and that every wave leaves, under random interleavings).
//
// Blocks are handed out in GROUPS of QB consecutive blocks, one group at a time per workgroup, groups in global order:
// at any moment the chip works on one compact window of the stream, and no wave holds work another one could do.
//   * level 1, workgroup (LDS): ONE 64-bit word holds the workgroup's current group, {global group, local group, next slot}.
//     A wave takes a block with a single returning add on that word: what comes back names the block completely, so there
//     is no window between "taking a slot" and "learning which block it is" (round 3; the round-2 queue read the group
//     from a ring after the take, and a wave stalled between the two could in principle be lapped by the ring -- the host
//     simulation produced exactly that).  The wave that takes the LAST slot of a group installs the next group's word;
//     waves that find the group used up wait for that (bounded) and take again.
//   * level 2, global: the wave that takes slot 0 of local group g draws the global group of local group g + Q_AHEAD with
//     one returning atomic and publishes it in a small look-ahead ring (read once, by the installer of that group).  Local
//     groups 0 and 1 are static (workgroup b: global groups b and wgs + b); group 0 is also TAKEN statically (wave w =
//     slot w, so the first block's rows can be requested before the tables are copied): the word starts at group 1 and wave
//     0 owes the fetch for group Q_AHEAD (queue_start).
// Tickets are drawn IN LOCAL ORDER: the fetch for local group i first waits until entry i - 1 is published, so a
// workgroup's global groups increase with the local group and the first group that reaches past the end of the stream ends
// it for the workgroup: a wave that draws a block >= nblocks leaves at once, and a wave that finds a used-up group leaves
// when that group already reaches the end.  Every wait is bounded: after Q_SPIN_LIMIT polls a wave leaves and counts a
// fault (third word of the queue block) instead of spinning for ever.
// TAIL (round 3, short launches only: queue_main_blocks): with two waves per SIMD a launch is a whole number of two-wave rounds
// plus a remainder.  When the remainder is at most one block per SIMD the launcher keeps it out of the groups (nblocks_main < nblocks): after the last group every
// SIMD's FIRST wave to arrive takes one tail block and the second wave leaves, so the last round runs one wave per SIMD
// (a block then takes ~0.6 of its two-wave time) instead of two waves on some SIMDs and none on others.  Tail blocks are
// handed out four at a time (one per SIMD of a workgroup) through a second global counter, first come first served: the
// workgroups that finish their groups first get them.
// One global atomic per QB blocks (a single address takes ~88 atomics/us; 70 k blocks in 0.5 ms would be 140/us), two
// groups of slack before anybody needs its result.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define IF_FIR_Q_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define IF_FIR_Q_FN inline __attribute__((always_inline))
#endif

namespace if_fir
{

constexpr unsigned QB = 8;        // blocks per group = waves per workgroup
constexpr unsigned Q_AHEAD = 2;   // groups fetched ahead; local groups 0 .. Q_AHEAD-1 are static
constexpr unsigned Q_RING = 4;    // look-ahead ring entries (a power of two >= Q_AHEAD + 2)
constexpr unsigned Q_SPIN_LIMIT = 1u << 22;
constexpr int64_t Q_NONE = (int64_t)1 << 46; // "no block" (beyond any stream, small enough to be multiplied by a block length)

// current-group word: [63:32] global group, [31:8] local group (24 bits), [7:0] next slot
IF_FIR_Q_FN unsigned long long queue_word(unsigned global_group, unsigned local_group, unsigned slot)
{
    return ((unsigned long long)global_group << 32) | ((unsigned long long)(local_group & 0xffffffu) << 8) | slot;
}
// look-ahead ring entry: [63:32] global group, [31:0] local group
IF_FIR_Q_FN unsigned long long queue_entry(unsigned global_group, unsigned local_group)
{
    return ((unsigned long long)global_group << 32) | local_group;
}

// P (platform) provides, wave-uniformly:
//   unsigned long long cur_add()              returning add of 1 on the current-group word (LDS)
//   unsigned long long cur_load()
//   void cur_store(unsigned long long)
//   unsigned long long ring_load(unsigned i)  look-ahead ring (LDS)
//   void ring_store(unsigned i, unsigned long long v)
//   unsigned ticket()                         global fetch-and-increment of the launch's ticket counter
//   void fault()                              count a bounded-wait expiry
//   void pause()                              back off inside a wait loop
//   unsigned wgs()                            workgroups of the launch
//   unsigned tail_claim(unsigned simd)        LDS fetch-and-increment of the SIMD's tail claim counter
//   unsigned long long tail_add()             LDS returning add of 1 on the workgroup's tail word
//   unsigned long long tail_load() / void tail_store(v)
//   unsigned tail_ticket()                    global fetch-and-increment of the launch's tail counter

template <class P>
IF_FIR_Q_FN bool queue_wait_entry(P &p, unsigned i, unsigned long long &e)
{
    for (unsigned spin = 0; spin < Q_SPIN_LIMIT; spin++)
    {
        e = p.ring_load(i & (Q_RING - 1));
        if ((unsigned)e == i)
            return true;
        p.pause();
    }
    p.fault();
    return false;
}

// draw the global group of local group i (in local order) and publish it
template <class P>
IF_FIR_Q_FN void queue_fetch(P &p, unsigned i)
{
    unsigned long long prev;
    if (!queue_wait_entry(p, i - 1, prev)) // (entries 0 .. Q_AHEAD-1 exist from the start)
        return;
    const unsigned t = p.ticket();
    p.ring_store(i & (Q_RING - 1), queue_entry(Q_AHEAD * p.wgs() + t, i));
}

// the fetch the static slot 0 of local group 0 (wave 0) owes: once per workgroup, after the LDS words are initialised
template <class P>
IF_FIR_Q_FN void queue_start(P &p)
{
    queue_fetch(p, Q_AHEAD);
}

// tail phase: at most one block per SIMD (see the header comment); tail word: 0 = nobody here yet, low byte = arrivals,
// [63:32] = 1 + first tail block of this workgroup once the first arrival has drawn it
template <class P>
IF_FIR_Q_FN int64_t queue_tail(P &p, unsigned simd, int64_t nblocks_main, int64_t nblocks)
{
    if (p.tail_claim(simd) != 0)
        return Q_NONE; // this SIMD's first wave has (or had) its tail block: the last round runs one wave per SIMD
    unsigned long long w = p.tail_add();
    if ((w & 0xffu) == 0)
    {
        const unsigned t = p.tail_ticket();
        w = ((unsigned long long)(4u * t + 1u) << 32) | 1u;
        p.tail_store(w);
    }
    else
    {
        unsigned spin = 0;
        for (; spin < Q_SPIN_LIMIT && (w >> 32) == 0; spin++)
        {
            p.pause();
            w = p.tail_load();
        }
        if ((w >> 32) == 0)
        {
            p.fault();
            return Q_NONE;
        }
    }
    const int64_t blk = nblocks_main + (int64_t)((unsigned)(w >> 32) - 1u) + simd;
    return blk < nblocks ? blk : Q_NONE;
}

// next block of this wave (>= nblocks: leave); *local_group (optional) receives the local group of a block of the groups.
// nblocks_main (a multiple of QB when < nblocks) = the blocks handed out in groups; the rest is the tail.
template <class P>
IF_FIR_Q_FN int64_t queue_take(P &p, unsigned simd, int64_t nblocks_main, int64_t nblocks, unsigned *local_group = nullptr)
{
    for (unsigned spin = 0; spin < Q_SPIN_LIMIT; spin++)
    {
        const unsigned long long w = p.cur_add();
        const unsigned gg = (unsigned)(w >> 32), g = (unsigned)(w >> 8) & 0xffffffu, j = (unsigned)w & 0xffu;
        if (j < QB)
        {
            if (j == 0)
                queue_fetch(p, g + Q_AHEAD);
            if (j == QB - 1)
            {
                // last slot: install the next group (its entry was fetched Q_AHEAD groups ago)
                unsigned long long e;
                if (queue_wait_entry(p, g + 1, e))
                    p.cur_store(queue_word((unsigned)(e >> 32), g + 1, 0));
            }
            if (local_group)
                *local_group = g;
            const int64_t blk = (int64_t)gg * QB + j;
            if (blk < nblocks_main)
                return blk;
            return nblocks_main < nblocks ? queue_tail(p, simd, nblocks_main, nblocks) : Q_NONE;
        }
        // the group is used up.  If it already reaches the end of the groups no later group is in range (ordered tickets)
        if ((int64_t)gg * QB + QB - 1 >= nblocks_main)
            return nblocks_main < nblocks ? queue_tail(p, simd, nblocks_main, nblocks) : Q_NONE;
        // otherwise wait (without adding again) until the taker of its last slot has installed the next group
        for (; spin < Q_SPIN_LIMIT; spin++)
        {
            const unsigned long long c = p.cur_load();
            if (((unsigned)(c >> 8) & 0xffffffu) != g)
                break;
            p.pause();
        }
    }
    p.fault();
    return Q_NONE;
}

// The launcher's split of a launch into groups and tail for `wgs` workgroups (4 SIMDs, two waves per SIMD each): returns
// nblocks_main.  p = whole two-wave rounds; the remainder goes to the tail only if it is at most one block per SIMD, and only
// in SHORT launches (1 <= p <= Q_TAIL_MAX_ROUNDS): measured on MI355X (profiles/r03_queue_tail_ab.txt) the tail shortens a
// 2^24-sample call by 14 % (2.13 blocks per wave) and a 2^25-sample one by 6 %, does nothing from 8 rounds on, and costs
// 1.5 % at 34 rounds, where the spread between fast and slow waves is larger than the remainder.
constexpr int64_t Q_TAIL_MAX_ROUNDS = 6;
IF_FIR_Q_FN int64_t queue_main_blocks(int64_t nblocks, int64_t wgs, int64_t max_rounds = Q_TAIL_MAX_ROUNDS)
{
    const int64_t simds = 4 * wgs, p = nblocks / (2 * simds), rem = nblocks - 2 * p * simds;
    return (p >= 1 && p <= max_rounds && rem > 0 && rem <= simds) ? 2 * p * simds : nblocks;
}

// initial LDS image of workgroup `wg`: the current-group word (static_first: local group 1 = global group wgs + wg, group 0
// being taken statically; else local group 0) and look-ahead ring entry i
IF_FIR_Q_FN unsigned long long queue_cur_init(unsigned wg, unsigned wgs, bool static_first)
{
    return static_first ? queue_word(wgs + wg, 1, 0) : queue_word(wg, 0, 0);
}
IF_FIR_Q_FN unsigned long long queue_ring_init(unsigned i, unsigned wg, unsigned wgs)
{
    return i < Q_AHEAD ? queue_entry(i * wgs + wg, i) : ~0ull;
}

} // namespace if_fir
