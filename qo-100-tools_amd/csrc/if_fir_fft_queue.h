// if_fir_fft_queue.h — the two-level block queue of the overlap-save kernel, written once for the device and for a host
// simulation (tests/c/fft_queue_sim.cpp runs the very same code with threads as waves and checks that every block is
// handed out exactly once and that every wave leaves, under random interleavings).
//
// Blocks are handed out in GROUPS of QB consecutive blocks, one group at a time per workgroup, groups in global order:
// at any moment the chip works on one compact window of the stream, and no wave holds work another one could do.
//   * level 1, workgroup (LDS): a wave takes the next SLOT of its workgroup: slot s = block s % QB of local group s / QB;
//   * level 2, global: the wave that takes slot 0 of local group g draws the global group of local group g + Q_AHEAD with
//     one returning atomic and publishes it in a ring of {local group, global group} words; local groups 0 .. Q_AHEAD-1
//     are static (workgroup b: global groups b, wgs + b).
// Round 3:
//   * tickets are drawn IN LOCAL ORDER: the wave that fetches for local group i first waits until entry i-1 is published
//     (its ticket has then been drawn), so a workgroup's global groups increase with the local group.  The first local
//     group that is out of range therefore ends the stream for its workgroup: a wave that draws a block >= nblocks may
//     leave at once without stranding an in-range group published for a later local group;
//   * the first slot of every wave is static (wave w = slot w of local group 0), so that the first block's rows can be
//     requested before the tables are copied; wave 0 owes the fetch for local group Q_AHEAD (queue_start);
//   * every wait is bounded: a wave that does not see its ring entry after Q_SPIN_LIMIT polls leaves (and counts a fault
//     in the queue block's third word) instead of spinning for ever.
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
constexpr unsigned Q_AHEAD = 2;   // groups fetched ahead = static groups per workgroup
constexpr unsigned Q_RING = 16;   // ring entries (a power of two, > Q_AHEAD + 1)
constexpr unsigned Q_SPIN_LIMIT = 1u << 22;
constexpr int64_t Q_NONE = (int64_t)1 << 46; // "no block" (beyond any stream, small enough to be multiplied by a block length)

// P (platform) provides, wave-uniformly:
//   unsigned slot_add()                       LDS fetch-and-increment of the workgroup's slot counter
//   unsigned long long ring_load(unsigned i)  ring[i]
//   void ring_store(unsigned i, unsigned long long v)
//   unsigned ticket()                         global fetch-and-increment of the launch's ticket counter
//   void fault()                              count a bounded-wait expiry
//   void pause()                              back off inside a wait loop
//   unsigned wgs()                            workgroups of the launch

template <class P>
IF_FIR_Q_FN bool queue_wait_entry(P &p, unsigned g, unsigned long long &e)
{
    for (unsigned spin = 0; spin < Q_SPIN_LIMIT; spin++)
    {
        e = p.ring_load(g & (Q_RING - 1));
        if ((unsigned)e == g)
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
    if (!queue_wait_entry(p, i - 1, prev)) // entries 0 .. Q_AHEAD-1 exist from the start, so i - 1 >= Q_AHEAD - 1 is defined
        return;
    const unsigned t = p.ticket();
    p.ring_store(i & (Q_RING - 1), ((unsigned long long)(Q_AHEAD * p.wgs() + t) << 32) | (unsigned long long)i);
}

// the fetch the static first slot 0 (wave 0) owes: call once per workgroup after the ring has been initialised
template <class P>
IF_FIR_Q_FN void queue_start(P &p)
{
    queue_fetch(p, Q_AHEAD);
}

// next block of this wave (Q_NONE: leave); *local_group (optional) receives the local group of the slot
template <class P>
IF_FIR_Q_FN int64_t queue_take(P &p, unsigned *local_group = nullptr)
{
    const unsigned s = p.slot_add();
    const unsigned g = s / QB, j = s % QB;
    if (j == 0)
        queue_fetch(p, g + Q_AHEAD);
    unsigned long long e;
    if (!queue_wait_entry(p, g, e))
        return Q_NONE;
    if (local_group)
        *local_group = g;
    return (int64_t)(unsigned)(e >> 32) * QB + j;
}

// initial ring image of workgroup `wg`: entry i (i < Q_RING)
IF_FIR_Q_FN unsigned long long queue_ring_init(unsigned i, unsigned wg, unsigned wgs)
{
    return i < Q_AHEAD ? (((unsigned long long)(i * wgs + wg)) << 32) | i : ~0ull;
}

} // namespace if_fir
