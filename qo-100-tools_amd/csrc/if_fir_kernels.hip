// if_fir_kernels.hip — hand-written CDNA4 (gfx950) kernels of the IF-chain FIR path.
//
// BUILD-DEFINED path: the reference vankxr/qo-100-tools holds no filter code
// (/root/reference/util/if-bandpass-filter/schematic.svg:174-222 is an analog LC drawing), so there is no
// reference kernel these follow; semantics are docs/SPEC.md, checked against oracle/ by tests/.
//
// Direct-form kernel (`fir_direct_kernel`) — "sample-stationary" register blocking:
//   * a workgroup stages one tile of interleaved I/Q (+ a T-1 halo) from HBM into LDS with coalesced 16-byte loads;
//   * every lane owns R consecutive (decimated) outputs; it walks the input samples its outputs depend on
//     ONCE, oldest first, reading two complex samples per ds_read_b128, and applies each sample to all R
//     accumulators with v_pk_fma_f32 (I and Q in one packed FMA, the tap broadcast from an SGPR via op_sel);
//   * the loop is fully unrolled over the (compile-time) tap count so every tap index is a constant: taps are
//     fetched with s_load into SGPRs, edge taps cost nothing, and accumulation segments (SPEC §3) are free of
//     run-time tests;
//   * LDS layout: per-lane chunks of D·R samples, each followed by 16 pad bytes, so that the 64 lanes of a
//     ds_read_b128 (stride D·R·8+16 bytes) fall on distinct bank groups and every address is lane-base + immediate.
//   * no MFMA: this is a 1-D convolution; the packed-FP32 VALU and HBM are the two rooflines.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "if_fir_kernels.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#include "generated/if_fir_walk_gen.h"

namespace if_fir
{

// --------------------------------------------------------------------------------------------------------------
// compile-time geometry of one direct-form instantiation
// --------------------------------------------------------------------------------------------------------------
template <int T, int D, int R, int BLOCK>
struct Geo
{
    static constexpr int DR = D * R;                               // input samples per lane chunk
    static constexpr int HALO = ((T - 1 + DR - 1) / DR) * DR;      // halo rounded up to whole chunks
    static constexpr int TILE_OUT = BLOCK * R;                     // outputs per tile
    static constexpr int TILE_IN = TILE_OUT * D;                   // new input samples per tile
    static constexpr int CHUNK_BYTES = DR * 8 + 16;                // padded chunk stride
    static constexpr int N_CHUNKS = BLOCK + HALO / DR;
    static constexpr int LDS_IN_BYTES = N_CHUNKS * CHUNK_BYTES;
    static constexpr int OCHUNK_BYTES = R * 8 + 16;                // padded output chunk (LDS-staged stores)
    static constexpr int LDS_OUT_BYTES = BLOCK * OCHUNK_BYTES;
    static constexpr int LDS_BYTES = LDS_IN_BYTES > LDS_OUT_BYTES ? LDS_IN_BYTES : LDS_OUT_BYTES;
    static_assert(DR % 4 == 0, "chunk must be a multiple of 4 samples (bank-conflict-free padding rule)");
    static_assert((T - 1) % 2 == 0, "odd tap counts only (pairs of samples are 16-byte aligned in LDS)");
};

// int16 IQ front-end: one dword = (I, Q) as two int16, value = int16 * 2^-15 (exact)
__device__ __forceinline__ f2 load_i16(const f2 *__restrict__ in, int64_t g)
{
    const int w = reinterpret_cast<const int *>(in)[g];
    return (f2){(float)(short)(w & 0xffff) * 0x1p-15f, (float)(w >> 16) * 0x1p-15f};
}

template <bool I16 = false>
__device__ __forceinline__ f2 fetch_sample(const f2 *__restrict__ in, const f2 *__restrict__ hist, int T, int64_t g,
                                           int64_t N)
{
    if (g >= 0)
        return (g < N) ? (I16 ? load_i16(in, g) : in[g]) : (f2){0.f, 0.f};
    const int64_t h = (int64_t)(T - 1) + g;
    // the history of an int16 stream is kept as raw int16 pairs as well
    return (h >= 0) ? (I16 ? load_i16(hist, h) : hist[h]) : (f2){0.f, 0.f};
}

// one input sample applied to all R accumulators.  C = sample offset relative to the lane's first output sample.
template <int T, int D, int R, int SEG, int C>
__device__ __forceinline__ void apply_sample(const f2 s, f2 (&acc)[R], f2 (&tot)[R], const float *__restrict__ taps)
{
#pragma unroll
    for (int r = 0; r < R; r++)
    {
        constexpr int dummy = 0;
        (void)dummy;
        const int k = D * r - C;
        if (k >= 0 && k < T)
        {
            const float h = taps[k];
            const f2 hh = {h, h};
            const bool first = (k == T - 1) || (k % SEG == SEG - 1);
            acc[r] = __builtin_elementwise_fma(s, hh, first ? (f2){0.f, 0.f} : acc[r]);
            if (k % SEG == 0)
                tot[r] = (k / SEG == (T - 1) / SEG) ? acc[r] : tot[r] + acc[r];
        }
    }
}

template <int T, int D, int R, int SEG, int C0, int CEND>
struct Walk
{
    // C0 is even relative to -(T-1): one ds_read_b128 = samples C0 and C0+1
    __device__ static __forceinline__ void run(const char *lane_lds, f2 (&acc)[R], f2 (&tot)[R],
                                               const float *__restrict__ taps)
    {
        using G = Geo<T, D, R, 1>;
        constexpr int U = C0 + G::HALO;                                   // index inside the lane-relative window
        constexpr int OFF = (U / G::DR) * G::CHUNK_BYTES + (U % G::DR) * 8; // compile-time LDS byte offset
        const f4 v = *reinterpret_cast<const f4 *>(lane_lds + OFF);
        apply_sample<T, D, R, SEG, C0>((f2){v.x, v.y}, acc, tot, taps);
        if constexpr (C0 + 1 <= CEND)
            apply_sample<T, D, R, SEG, C0 + 1>((f2){v.z, v.w}, acc, tot, taps);
        if constexpr (C0 + 2 <= CEND)
            Walk<T, D, R, SEG, C0 + 2, CEND>::run(lane_lds, acc, tot, taps);
    }
};

// --------------------------------------------------------------------------------------------------------------
// direct-form kernel: one tile per workgroup
// --------------------------------------------------------------------------------------------------------------
template <int T, int D, int R, int SEG, int BLOCK, bool LDS_OUT>
__global__ __launch_bounds__(BLOCK) void fir_direct_kernel(const f2 *__restrict__ in, f2 *__restrict__ out,
                                                          const float *__restrict__ taps,
                                                          const f2 *__restrict__ hist, int64_t N, int32_t n0,
                                                          int64_t M)
{
    using G = Geo<T, D, R, BLOCK>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t tile_base = (int64_t)n0 + tile * G::TILE_IN; // stream-relative index of the tile's first output sample
    const int64_t g0 = tile_base - G::HALO;                   // first sample staged in LDS

    // ---- stage tile + halo into LDS (padded chunk layout) ------------------------------------------------------
    constexpr int PAIRS = (G::TILE_IN + G::HALO) / 2; // 16-byte units
    const bool interior = (g0 >= 0) && (g0 + G::TILE_IN + G::HALO <= N) && ((g0 & 1) == 0);
    if (interior)
    {
        const f4 *src = reinterpret_cast<const f4 *>(in + g0);
#pragma unroll
        for (int i = 0; i < (PAIRS + BLOCK - 1) / BLOCK; i++)
        {
            const int p = tid + i * BLOCK;
            if ((i + 1) * BLOCK <= PAIRS || p < PAIRS)
            {
                const f4 v = src[p];
                const int u = 2 * p;
                *reinterpret_cast<f4 *>(smem + (u / G::DR) * G::CHUNK_BYTES + (u % G::DR) * 8) = v;
            }
        }
    }
    else
    {
        for (int p = tid; p < PAIRS; p += BLOCK)
        {
            const int u = 2 * p;
            const f2 a = fetch_sample(in, hist, T, g0 + u, N);
            const f2 b = fetch_sample(in, hist, T, g0 + u + 1, N);
            *reinterpret_cast<f4 *>(smem + (u / G::DR) * G::CHUNK_BYTES + (u % G::DR) * 8) = (f4){a.x, a.y, b.x, b.y};
        }
    }
    __syncthreads();

    // ---- compute: walk the samples, oldest first ----------------------------------------------------------------
    f2 acc[R], tot[R];
#pragma unroll
    for (int r = 0; r < R; r++)
    {
        acc[r] = (f2){0.f, 0.f};
        tot[r] = (f2){0.f, 0.f};
    }
    const char *lane_lds = smem + tid * G::CHUNK_BYTES;
    Walk<T, D, R, SEG, -(T - 1), D *(R - 1)>::run(lane_lds, acc, tot, taps);

    // ---- store ----------------------------------------------------------------------------------------------------
    const int64_t m_tile = tile * G::TILE_OUT;
    if constexpr (LDS_OUT)
    {
        __syncthreads(); // everyone is done reading the input tile
        char *o = smem + tid * G::OCHUNK_BYTES;
#pragma unroll
        for (int r = 0; r < R; r += 2)
            *reinterpret_cast<f4 *>(o + r * 8) = (f4){tot[r].x, tot[r].y, tot[r + 1].x, tot[r + 1].y};
        __syncthreads();
        constexpr int OPAIRS = G::TILE_OUT / 2;
        const bool full = (m_tile + G::TILE_OUT <= M);
#pragma unroll
        for (int i = 0; i < OPAIRS / BLOCK; i++)
        {
            const int p = tid + i * BLOCK; // pair index inside the tile
            const int e = 2 * p;           // output index inside the tile
            const f4 v = *reinterpret_cast<const f4 *>(smem + (e / R) * G::OCHUNK_BYTES + (e % R) * 8);
            if (full)
                *reinterpret_cast<f4 *>(out + m_tile + e) = v;
            else
            {
                if (m_tile + e < M)
                    out[m_tile + e] = (f2){v.x, v.y};
                if (m_tile + e + 1 < M)
                    out[m_tile + e + 1] = (f2){v.z, v.w};
            }
        }
    }
    else
    {
        const int64_t m_lane = m_tile + (int64_t)tid * R;
        if (m_lane + R <= M)
        {
#pragma unroll
            for (int r = 0; r < R; r += 2)
                *reinterpret_cast<f4 *>(out + m_lane + r) = (f4){tot[r].x, tot[r].y, tot[r + 1].x, tot[r + 1].y};
        }
        else
        {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (m_lane + r < M)
                    out[m_lane + r] = tot[r];
        }
    }
}

// --------------------------------------------------------------------------------------------------------------
// direct-form kernel v2 — persistent, wave-private pipeline (no workgroup barriers)
//
// Each WAVE owns a private LDS window (halo + one tile of 64·R outputs) and walks a contiguous run of tiles:
//   issue the global loads of tile i+1 into registers  →  walk tile i out of LDS (the FMA stream)  →
//   move the last T-1 samples to the front of the window (halo carry: the stream is read from HBM exactly once)  →
//   transpose the 64·R outputs through the dead part of the window and store them with 1 KiB-contiguous
//   stores  →  write tile i+1 from registers into the window.
// Waves never synchronise with each other, so the two waves that share a SIMD drift apart and one computes
// while the other moves data.  A 256-thread workgroup is four such waves (one per SIMD); two workgroups fit a CU.
// --------------------------------------------------------------------------------------------------------------
template <int T, int D, int R>
struct WGeo
{
    static constexpr int DR = D * R;
    static constexpr int HALO = ((T - 1 + DR - 1) / DR) * DR;
    static constexpr int HCH = HALO / DR;                 // halo chunks
    static constexpr int CH = DR * 8 + 16;                // padded chunk stride (bytes)
    static constexpr int TILE_OUT = 64 * R;
    static constexpr int TILE_IN = 64 * DR;
    static constexpr int NCH = 64 + HCH;
    static constexpr int WAVE_LDS = NCH * CH;
    static constexpr int OCH = R * 8 + 16;                // padded per-lane output chunk
    static constexpr int OUT_OFF = HCH * CH;              // outputs are transposed through the (dead) tile area
    static constexpr int TILE_UNITS = TILE_IN / 2;        // 16-byte units in one tile
    static constexpr int LOADS = TILE_UNITS / 64;         // dwordx4 loads per lane per tile
    static constexpr int HALO_UNITS = HCH * CH / 16;      // halo carry copies whole padded chunks
    static_assert(DR % 4 == 0 && (T - 1) % 2 == 0, "see Geo");
    // the outputs are staged after the halo carry, when the whole tile area [OUT_OFF, WAVE_LDS) is dead
    static_assert(OUT_OFF + 64 * OCH <= WAVE_LDS, "output transposition must fit in the dead tile area");
    static_assert(TILE_UNITS % 64 == 0 && (DR % 2) == 0, "tile must be a whole number of wave-wide 16-byte loads");
};

// LDS byte address of 16-byte unit p (= samples 2p, 2p+1) of the tile area, padded chunk layout
template <typename G>
__device__ __forceinline__ int tile_unit_addr(int p)
{
    const int u = 2 * p;
    return (G::HCH + u / G::DR) * G::CH + (u % G::DR) * 8;
}

// selects one generated assembly walk (tools/gen_walk.py): V = schedule variant of the same arithmetic
template <int T, int D, int R, int SEG, int V>
struct AsmWalk;
#define IF_FIR_ASM_WALK(T_, D_, R_, S_, V_, FN_)                                                       \
    template <>                                                                                        \
    struct AsmWalk<T_, D_, R_, S_, V_>                                                                 \
    {                                                                                                  \
        static __device__ __forceinline__ void run(unsigned a, const float *t, f2 (&tot)[R_]) { FN_(a, t, tot); } \
    };
IF_FIR_ASM_WALK(255, 4, 8, 32, 0, walk_asm_T255_D4_R8_S32)
IF_FIR_ASM_WALK(255, 4, 8, 32, 2, walk_asm_T255_D4_R8_S32_b128)
IF_FIR_ASM_WALK(127, 4, 8, 32, 0, walk_asm_T127_D4_R8_S32)

typedef __attribute__((address_space(3))) char lds_char_t;

// fill `count` 16-byte units of the wave's window starting at LDS unit index `unit0` (padded chunk layout, unit 0 =
// first halo sample pair) from stream sample g_first on, with bounds/history checks (slow path: edges only)
template <typename G>
__device__ __forceinline__ void fill_slow(char *wl, int lane, int unit0, int count, const f2 *__restrict__ in,
                                          const f2 *__restrict__ hist, int T, int64_t g_first, int64_t N)
{
    for (int p = lane; p < count; p += 64)
    {
        const int u = 2 * (unit0 + p);
        const f2 a = fetch_sample(in, hist, T, g_first + 2 * p, N);
        const f2 b = fetch_sample(in, hist, T, g_first + 2 * p + 1, N);
        *reinterpret_cast<f4 *>(wl + (u / G::DR) * G::CH + (u % G::DR) * 8) = (f4){a.x, a.y, b.x, b.y};
    }
}

template <int T, int D, int R, int SEG, int V, int WPS>
__global__ __launch_bounds__(256, WPS) void fir_direct_wave_kernel(const f2 *__restrict__ in, f2 *__restrict__ out,
                                                                  const float *__restrict__ taps,
                                                                  const f2 *__restrict__ hist, int64_t N, int32_t n0,
                                                                  int64_t M, int64_t tiles_total, int32_t run_len,
                                                                  int32_t n_long, int32_t waves_total,
                                                                  unsigned int *queue, unsigned long long *dbg)
{
    using G = WGeo<T, D, R>;
    constexpr int HU = G::HALO / 2;                        // 16-byte units in the halo
    constexpr int HL = (HU + 63) / 64;                     // halo loads per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // diagnostic stamps (only when the host passes a buffer; they go nowhere else): start/end of this wave
    const unsigned long long st_real = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long st_clk = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wl = smem + wid * G::WAVE_LDS;                   // this wave's private window
    // 32-bit LDS byte address of this lane's chunk (operand of the generated ds_read instructions)
    const unsigned lds_lane_addr = (unsigned)(uintptr_t)(lds_char_t *)smem + (unsigned)(wid * G::WAVE_LDS + lane * G::CH);
    const int64_t gw = (int64_t)blockIdx.x * 4 + wid;
    const bool aligned = ((n0 & 1) == 0);
    // guided schedule: tickets [0, n_long) are runs of run_len tiles, later tickets single tiles (short tail)
    const int64_t runs_total = (int64_t)n_long + (tiles_total - (int64_t)n_long * run_len);

    // Work distribution: run r = tiles [r*run_len, (r+1)*run_len).  Wave gw starts with run gw; further runs come from
    // an atomic queue, always grabbed one run ahead so the atomic's latency hides behind a whole run.  (The two
    // waves sharing a SIMD do not progress at the same speed — the older one wins issue arbitration — so a static
    // split leaves half the chip idle at the end.)
    int64_t run = gw;
    unsigned int ticket = 0;
    bool have_ticket = false;
    if (run < runs_total)
    {
        // ---- prologue: halo + first tile of the first run (bounds-checked path; once per wave) -----------------
        fill_slow<G>(wl, lane, 0, HU + G::TILE_UNITS, in, hist, T,
                     (int64_t)n0 + ((run < n_long) ? run * run_len : (int64_t)n_long * run_len + (run - n_long)) * G::TILE_IN - G::HALO, N);
    }
    while (run < runs_total)
    {
        const int64_t t_first = (run < n_long) ? run * run_len : (int64_t)n_long * run_len + (run - n_long);
        const int64_t t_last = (run < n_long) ? t_first + run_len : t_first + 1;
        if (lane == 0)
            ticket = atomicAdd(queue, 1u);                  // reservation for the run after this one
        have_ticket = true;
        int64_t next_run = runs_total;                      // resolved when the current run reaches its last tile

        for (int64_t t = t_first; t < t_last; t++)
        {
            // ---- 0. which tile comes next? ----------------------------------------------------------------------
            bool contiguous = (t + 1 < t_last);
            int64_t t_next = t + 1;
            if (!contiguous)
            {
                next_run = (int64_t)waves_total + (int64_t)__builtin_amdgcn_readfirstlane(ticket);
                have_ticket = false;
                t_next = (next_run < n_long) ? next_run * run_len : (int64_t)n_long * run_len + (next_run - n_long);
            }
            const bool has_next = contiguous || (next_run < runs_total);
            const int64_t base_next = (int64_t)n0 + t_next * G::TILE_IN; // first sample of the next tile
            const bool next_fast = has_next && aligned && (base_next + G::TILE_IN <= N) &&
                                   (contiguous || base_next - G::HALO >= 0);

            // ---- 1. prefetch the next tile (and its halo when it starts a new run) into registers --------------
            f4 nxt[G::LOADS];
            f4 nh[HL];
            if (next_fast)
            {
                const f4 *src = reinterpret_cast<const f4 *>(in + base_next) + lane;
#pragma unroll
                for (int i = 0; i < G::LOADS; i++)
                    nxt[i] = src[i * 64];
                if (!contiguous)
                {
                    const f4 *hsrc = reinterpret_cast<const f4 *>(in + base_next - G::HALO) + lane;
#pragma unroll
                    for (int i = 0; i < HL; i++)
                        if ((i + 1) * 64 <= HU || lane + i * 64 < HU)
                            nh[i] = hsrc[i * 64];
                }
            }

            // ---- 2. walk tile t (generated assembly: FMA stream with LDS reads prefetched into a register ring) --
            f2 tot[R];
            AsmWalk<T, D, R, SEG, V>::run(lds_lane_addr, taps, tot);

            // ---- 3. halo for the next tile -----------------------------------------------------------------------
            if (contiguous)
            {
                // carry: last HCH chunks -> front of the window (the stream is read from HBM exactly once)
                f4 hv[(G::HALO_UNITS + 63) / 64];
#pragma unroll
                for (int i = 0; i < (G::HALO_UNITS + 63) / 64; i++)
                {
                    const int p = lane + i * 64;
                    if ((i + 1) * 64 <= G::HALO_UNITS || p < G::HALO_UNITS)
                        hv[i] = *reinterpret_cast<const f4 *>(wl + 64 * G::CH + p * 16);
                }
#pragma unroll
                for (int i = 0; i < (G::HALO_UNITS + 63) / 64; i++)
                {
                    const int p = lane + i * 64;
                    if ((i + 1) * 64 <= G::HALO_UNITS || p < G::HALO_UNITS)
                        *reinterpret_cast<f4 *>(wl + p * 16) = hv[i];
                }
            }
            else if (next_fast)
            {
#pragma unroll
                for (int i = 0; i < HL; i++)
                {
                    const int u = 2 * (lane + i * 64);
                    if ((i + 1) * 64 <= HU || lane + i * 64 < HU)
                        *reinterpret_cast<f4 *>(wl + (u / G::DR) * G::CH + (u % G::DR) * 8) = nh[i];
                }
            }

            // ---- 4. outputs: transpose through LDS, 1 KiB-contiguous stores -------------------------------------
            {
                char *o = wl + G::OUT_OFF + lane * G::OCH;
#pragma unroll
                for (int r = 0; r < R; r += 2)
                    *reinterpret_cast<f4 *>(o + r * 8) = (f4){tot[r].x, tot[r].y, tot[r + 1].x, tot[r + 1].y};
                const int64_t m_tile = t * G::TILE_OUT;
                const bool full = (m_tile + G::TILE_OUT <= M);
#pragma unroll
                for (int i = 0; i < R / 2; i++)
                {
                    const int e = 2 * (lane + 64 * i); // output index inside the tile
                    const f4 v = *reinterpret_cast<const f4 *>(wl + G::OUT_OFF + (e / R) * G::OCH + (e % R) * 8);
                    if (full)
                        __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out + m_tile + e)); // never read again
                    else
                    {
                        if (m_tile + e < M)
                            out[m_tile + e] = (f2){v.x, v.y};
                        if (m_tile + e + 1 < M)
                            out[m_tile + e + 1] = (f2){v.z, v.w};
                    }
                }
            }

            // ---- 5. next tile into the window ------------------------------------------------------------------
            if (next_fast)
            {
#pragma unroll
                for (int i = 0; i < G::LOADS; i++)
                    *reinterpret_cast<f4 *>(wl + tile_unit_addr<G>(lane + 64 * i)) = nxt[i];
            }
            else if (has_next)
            {
                if (contiguous)
                    fill_slow<G>(wl, lane, HU, G::TILE_UNITS, in, hist, T, base_next, N);
                else
                    fill_slow<G>(wl, lane, 0, HU + G::TILE_UNITS, in, hist, T, base_next - G::HALO, N);
            }
        }
        run = next_run;
    }
    (void)have_ticket;
    if (dbg && (threadIdx.x & 63) == 0)
    {
        const long long gwi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
        dbg[4 * gwi + 0] = st_real;
        dbg[4 * gwi + 1] = __builtin_amdgcn_s_memrealtime();
        dbg[4 * gwi + 2] = st_clk;
        dbg[4 * gwi + 3] = __builtin_amdgcn_s_memtime();
    }
}

// --------------------------------------------------------------------------------------------------------------
// tap-split kernel (the north_star's wording made literal): any T <= 4096, any D <= 64.
//   * the taps are staged in LDS, de-interleaved by k mod 4, so that lane q of every quad owns the taps 4j+q and
//     reads four of them per ds_read_b128;
//   * the input tile (+ T-1 halo) is staged in LDS with coalesced loads; the 4 lanes of a quad read 4 neighbouring
//     samples per step (bank-conflict free), each lane feeding R_TS = 4 accumulators (outputs 64 apart) so a tap
//     fetched from LDS is used four times;
//   * the four partial sums of a quad are reduced with two DPP butterfly steps (quad_perm xor 1, xor 2):
//     (s0 + s1) + (s2 + s3); lane 0 of the quad stores.
// Summation order = oracle mode 3 (SPEC §3): inside a lane descending j, segments of SEG steps.
// --------------------------------------------------------------------------------------------------------------
constexpr int TS_R = 4;         // outputs per lane
constexpr int TS_TILE = 256;    // outputs per workgroup (64 quads x 4)
constexpr int TS_MAX_LDS = 150 * 1024;

__device__ __forceinline__ float dpp_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
}

template <int SEG>
__global__ __launch_bounds__(256) void fir_tapsplit_kernel(const f2 *__restrict__ in, f2 *__restrict__ out,
                                                          const float *__restrict__ taps,
                                                          const f2 *__restrict__ hist, int T, int D, int64_t N,
                                                          int32_t n0, int64_t M, int32_t tile_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int J = (T + 3) / 4;                 // steps per lane
    const int J4 = (J + 3) & ~3;               // padded to whole b128 reads
    float *tl = reinterpret_cast<float *>(smem);                 // tl[q*J4 + j] = h[4j+q] (0 beyond T)
    f2 *xs = reinterpret_cast<f2 *>(smem + 16 * J4);             // staged samples
    for (int i = tid; i < 4 * J4; i += 256)
    {
        const int q = i / J4, j = i - q * J4, k = 4 * j + q;
        tl[i] = (j < J && k < T) ? taps[k] : 0.0f;
    }
    const int64_t m_t = (int64_t)blockIdx.x * tile_out;          // first output of the tile
    const int64_t s_lo = (int64_t)n0 + m_t * D - (T - 1);        // stream index of xs[0]
    const int count = (tile_out - 1) * D + T;
    for (int i = tid; i < count; i += 256)
        xs[i] = fetch_sample(in, hist, T, s_lo + i, N);
    __syncthreads();

    const int quad = tid >> 2, q = tid & 3;
    const int per_quad = tile_out / 64;        // 1..4 outputs per lane, 64 apart
    f2 acc[TS_R], tot[TS_R];
    bool have_tot = false;
    int pos[TS_R];
#pragma unroll
    for (int r = 0; r < TS_R; r++)
    {
        acc[r] = (f2){0.f, 0.f};
        tot[r] = (f2){0.f, 0.f};
        const int o = quad + 64 * (r < per_quad ? r : 0);        // unused slots recompute slot 0 (results dropped)
        pos[r] = o * D + (T - 1) - q;                            // xs index of this lane's sample for j = 0
    }
    const float *tq = tl + q * J4;
    for (int jb = J4 - 4; jb >= 0; jb -= 4)
    {
        const f4 h4 = *reinterpret_cast<const f4 *>(tq + jb);
        const float hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
        for (int u = 3; u >= 0; u--)
        {
            const int j = jb + u;
            if (j < J)
            {
                const float h = hh[u];
#pragma unroll
                for (int r = 0; r < TS_R; r++)
                {
                    const int xi = pos[r] - 4 * j;
                    const f2 x = (xi >= 0) ? xs[xi] : (f2){0.f, 0.f}; // 4j+q beyond T-1 only meets zero taps
                    acc[r] = __builtin_elementwise_fma(x, (f2){h, h}, acc[r]);
                }
                if (j % SEG == 0)
                {
#pragma unroll
                    for (int r = 0; r < TS_R; r++)
                    {
                        tot[r] = have_tot ? tot[r] + acc[r] : acc[r];
                        acc[r] = (f2){0.f, 0.f};
                    }
                    have_tot = true;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TS_R; r++)
    {
        f2 v = tot[r];
        v.x += dpp_xor1(v.x);
        v.y += dpp_xor1(v.y);
        v.x += dpp_xor2(v.x);
        v.y += dpp_xor2(v.y);
        const int64_t m = m_t + quad + 64 * r;
        if (q == 0 && r < per_quad && m < M)
            out[m] = v;
    }
}

// --------------------------------------------------------------------------------------------------------------
// generic kernel: any T ≤ 4096, any D ≤ 64.  One output per thread, taps read through the scalar cache
// (uniform index), samples straight from global/L2 (neighbouring lanes share lines).  Same summation order
// as the fast kernels (descending k, segments of SEG).  Correctness fallback, not a performance path.
// --------------------------------------------------------------------------------------------------------------
template <int SEG, bool CTAPS, bool I16>
__global__ __launch_bounds__(256) void fir_generic_kernel(const f2 *__restrict__ in, f2 *__restrict__ out,
                                                         const float *__restrict__ taps,
                                                         const f2 *__restrict__ hist, int T, int D, int64_t N,
                                                         int32_t n0, int64_t M, uint32_t nco_on, uint32_t nco_phi0,
                                                         uint32_t nco_delta)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M)
        return;
    const int64_t n = (int64_t)n0 + m * D;
    f2 acc = {0.f, 0.f}, tot = {0.f, 0.f};
    const int top = (T - 1) / SEG;
    for (int k = T - 1; k >= 0; k--)
    {
        const f2 s = fetch_sample<I16>(in, hist, T, n - k, N);
        const bool first = (k == T - 1) || (k % SEG == SEG - 1);
        if (first)
            acc = (f2){0.f, 0.f};
        if constexpr (CTAPS)
        {
            // complex tap (hr, hi): re = fma(-xi, hi, fma(xr, hr, re)), im = fma(xi, hr, fma(xr, hi, im))
            const float hr = taps[2 * k], hi = taps[2 * k + 1];
            acc = __builtin_elementwise_fma((f2){-s.y, s.y}, (f2){hi, hr},
                                            __builtin_elementwise_fma((f2){s.x, s.x}, (f2){hr, hi}, acc));
        }
        else
        {
            const float h = taps[k];
            acc = __builtin_elementwise_fma(s, (f2){h, h}, acc);
        }
        if (k % SEG == 0)
            tot = (k / SEG == top) ? acc : tot + acc;
    }
    if (nco_on)
    {
        // SPEC §3.2: the NCO ahead of the filter = complex taps (already in `taps`) + this rotation of the output
        const float2 w = nco_phasor(nco_phi0 + (uint32_t)m * nco_delta);
        tot = (f2){fmaf(tot.x, w.x, -tot.y * w.y), fmaf(tot.x, w.y, tot.y * w.x)};
    }
    out[m] = tot;
}

// --------------------------------------------------------------------------------------------------------------
// history update: hist_out = last T-1 samples of (hist_in ‖ in)
// --------------------------------------------------------------------------------------------------------------
template <bool I16>
__global__ void fir_history_kernel(const f2 *__restrict__ in, const f2 *__restrict__ hist_in, f2 *__restrict__ hist_out,
                                   int H, int64_t N)
{
    // hist_out = the last H samples of (hist_in || in), raw (int16 pairs stay int16 pairs)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H)
        return;
    const int64_t g = N - (int64_t)H + i, h = (int64_t)H + g;
    if constexpr (I16)
    {
        const int *src = reinterpret_cast<const int *>(in), *hsrc = reinterpret_cast<const int *>(hist_in);
        reinterpret_cast<int *>(hist_out)[i] = (g >= 0) ? src[g] : (h >= 0 ? hsrc[h] : 0);
    }
    else
        hist_out[i] = (g >= 0) ? in[g] : (h >= 0 ? hist_in[h] : (f2){0.f, 0.f});
}

// --------------------------------------------------------------------------------------------------------------
// SPEC §5 synthetic generator (bit-identical to oracle_synth_iq)
// --------------------------------------------------------------------------------------------------------------
struct ToneTable
{
    float v[10];
};

__device__ __forceinline__ uint64_t splitmix_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_kernel(f2 *__restrict__ iq, uint64_t first, uint64_t count, uint64_t seed,
                                                   ToneTable tone)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride)
    {
        const uint64_t n = first + i;
        const uint64_t z = splitmix_mix(seed + (n + 1) * 0x9E3779B97F4A7C15ULL);
        const float ui = ((float)(uint32_t)(z >> 40) * 0x1p-24f - 0.5f) * 0.5f;
        const float uq = ((float)(uint32_t)((z >> 16) & 0xFFFFFFu) * 0x1p-24f - 0.5f) * 0.5f;
        const uint32_t p = (uint32_t)(n % 5u);
        float ti = tone.v[0], tq = tone.v[1];
#pragma unroll
        for (int j = 1; j < 5; j++)
        {
            ti = (p == (uint32_t)j) ? tone.v[2 * j] : ti;
            tq = (p == (uint32_t)j) ? tone.v[2 * j + 1] : tq;
        }
        iq[i] = (f2){ti + ui, tq + uq};
    }
}

// --------------------------------------------------------------------------------------------------------------
// host-side launchers (called from the C-ABI shim)
// --------------------------------------------------------------------------------------------------------------
template <int T, int D, int R, int SEG, int BLOCK, bool LDS_OUT>
static hipError_t launch_direct(const LaunchArgs &a)
{
    using G = Geo<T, D, R, BLOCK>;
    auto kern = fir_direct_kernel<T, D, R, SEG, BLOCK, LDS_OUT>;
    static DeviceSetup setup;
    {
        const hipError_t e = device_setup(setup, a.device, reinterpret_cast<const void *>(kern), G::LDS_BYTES, nullptr);
        if (e != hipSuccess)
            return e;
    }
    const int64_t tiles = (a.M + G::TILE_OUT - 1) / G::TILE_OUT;
    if (tiles <= 0)
        return hipSuccess;
    if (tiles > 0x7fffffffLL)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(BLOCK), G::LDS_BYTES, a.stream,
                       reinterpret_cast<const f2 *>(a.in), reinterpret_cast<f2 *>(a.out), a.taps,
                       reinterpret_cast<const f2 *>(a.hist), a.N, a.n0, a.M);
    return hipGetLastError();
}

template <int T, int D, int R, int SEG, int V>
static hipError_t launch_wave(const LaunchArgs &a, int run_len_arg)
{
    using G = WGeo<T, D, R>;
    constexpr int LDS = 4 * G::WAVE_LDS;
    constexpr int BPC = (160 * 1024) / LDS;                 // workgroups (of 4 waves) per CU by LDS
    static_assert(BPC >= 1, "window too large");
    constexpr int WPS = BPC >= 3 ? 3 : BPC >= 2 ? 2 : 1;    // waves per SIMD the register budget is sized for
    auto kern = fir_direct_wave_kernel<T, D, R, SEG, V, WPS>;
    static DeviceSetup setup;
    int ncus = 0;
    {
        const hipError_t e = device_setup(setup, a.device, reinterpret_cast<const void *>(kern), LDS, &ncus);
        if (e != hipSuccess)
            return e;
    }
    const int64_t tiles = (a.M + G::TILE_OUT - 1) / G::TILE_OUT;
    if (tiles <= 0)
        return hipSuccess;
    int64_t blocks = (int64_t)ncus * WPS;
    int64_t waves = blocks * 4;
    int64_t run_len = run_len_arg > 0 ? run_len_arg : 4;
    const int64_t per_wave = (tiles + waves - 1) / waves;
    if (run_len > per_wave)
        run_len = per_wave;
    // ~85 % of the tiles in runs of run_len (and at least one full run per wave), the tail tile by tile
    int64_t n_long = (tiles * 85 / 100) / run_len;
    if (n_long < waves)
        n_long = (tiles / run_len < waves) ? tiles / run_len : waves;
    const int64_t runs = n_long + (tiles - n_long * run_len);
    if (runs < waves)
    {
        blocks = (runs + 3) / 4;
        waves = blocks * 4;
    }
    if (a.queue_valid)
        *a.queue_valid = false; // the overlap-save launcher keeps a running ticket base on the same counter
    hipError_t e = hipMemsetAsync(a.queue, 0, 16, a.stream); // run queue: re-zeroed before every launch (word 4 = fault count, kept)
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), LDS, a.stream, reinterpret_cast<const f2 *>(a.in),
                       reinterpret_cast<f2 *>(a.out), a.taps, reinterpret_cast<const f2 *>(a.hist), a.N, a.n0, a.M,
                       tiles, (int32_t)run_len, (int32_t)n_long, (int32_t)waves, (unsigned int *)a.queue,
                       (unsigned long long *)a.dbg);
    return hipGetLastError();
}

hipError_t launch_fir(const LaunchArgs &a, int variant)
{
    if (a.backend == BACKEND_DIRECT)
    {
        // variant selects tuning alternatives of one (T, D) instantiation (bench/tests can sweep them)
        if (a.T == 255 && a.D == 4)
        {
            switch (variant)
            {
            case 1: return launch_direct<255, 4, 8, 32, 256, false>(a);
            case 2: return launch_direct<255, 4, 8, 32, 128, true>(a);
            case 3: return launch_direct<255, 4, 8, 32, 256, true>(a);
            case 4: return launch_wave<255, 4, 8, 32, 0>(a, 4);
            case 5: return launch_wave<255, 4, 8, 32, 2>(a, 8);
            case 6: return launch_wave<255, 4, 8, 32, 2>(a, 2);
            default: return launch_wave<255, 4, 8, 32, 2>(a, 4);
            }
        }
        if (a.T == 255 && a.D == 1)
        {
            switch (variant)
            {
            case 1: return launch_direct<255, 1, 16, 32, 256, false>(a);
            case 2: return launch_direct<255, 1, 8, 32, 256, true>(a);
            default: return launch_direct<255, 1, 16, 32, 256, true>(a);
            }
        }
        if (a.T == 127 && a.D == 1)
        {
            switch (variant)
            {
            case 1: return launch_direct<127, 1, 16, 32, 256, false>(a);
            case 2: return launch_direct<127, 1, 8, 32, 256, true>(a);
            default: return launch_direct<127, 1, 16, 32, 256, true>(a);
            }
        }
        if (a.T == 127 && a.D == 4)
        {
            switch (variant)
            {
            case 1: return launch_direct<127, 4, 8, 32, 256, true>(a);
            default: return launch_wave<127, 4, 8, 32, 0>(a, 4);
            }
        }
        return hipErrorInvalidConfiguration;
    }
    if (a.backend == BACKEND_TAPSPLIT)
    {
        if (a.M <= 0)
            return hipSuccess;
        // tile: as many outputs (64..256, multiple of 64) as keep taps + samples inside the LDS budget
        const int J4 = (((a.T + 3) / 4) + 3) & ~3;
        int tile = TS_TILE;
        while (tile > 64 && 16 * J4 + 8 * ((int64_t)(tile - 1) * a.D + a.T) > TS_MAX_LDS)
            tile -= 64;
        const int64_t lds = 16 * J4 + 8 * ((int64_t)(tile - 1) * a.D + a.T);
        if (lds > TS_MAX_LDS)
            return hipErrorInvalidConfiguration;
        auto kern = fir_tapsplit_kernel<32>;
        static DeviceSetup setup;
        {
            const hipError_t e = device_setup(setup, a.device, reinterpret_cast<const void *>(kern), TS_MAX_LDS, nullptr);
            if (e != hipSuccess)
                return e;
        }
        const int64_t blocks = (a.M + tile - 1) / tile;
        if (blocks > 0x7fffffffLL)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), (size_t)lds, a.stream,
                           reinterpret_cast<const f2 *>(a.in), reinterpret_cast<f2 *>(a.out), a.taps,
                           reinterpret_cast<const f2 *>(a.hist), a.T, a.D, a.N, a.n0, a.M, (int32_t)tile);
        return hipGetLastError();
    }
    if (a.backend == BACKEND_GENERIC)
    {
        if (a.M <= 0)
            return hipSuccess;
        const int64_t blocks = (a.M + 255) / 256;
        if (blocks > 0x7fffffffLL)
            return hipErrorInvalidValue;
#define IF_FIR_GENERIC_LAUNCH(CT_, I16_)                                                                          \
    hipLaunchKernelGGL((fir_generic_kernel<32, CT_, I16_>), dim3((unsigned)blocks), dim3(256), 0, a.stream,        \
                       reinterpret_cast<const f2 *>(a.in), reinterpret_cast<f2 *>(a.out), a.taps,                  \
                       reinterpret_cast<const f2 *>(a.hist), a.T, a.D, a.N, a.n0, a.M, a.nco_word, nco_phi0(a),   \
                       nco_delta(a))
        if (a.ctaps && a.in_i16)
            IF_FIR_GENERIC_LAUNCH(true, true);
        else if (a.ctaps)
            IF_FIR_GENERIC_LAUNCH(true, false);
        else if (a.in_i16)
            IF_FIR_GENERIC_LAUNCH(false, true);
        else
            IF_FIR_GENERIC_LAUNCH(false, false);
#undef IF_FIR_GENERIC_LAUNCH
        return hipGetLastError();
    }
    return hipErrorInvalidConfiguration;
}

bool direct_supported(int T, int D)
{
    return (T == 255 || T == 127) && (D == 1 || D == 4);
}

hipError_t launch_history(const void *in, const void *hist_in, void *hist_out, int hist_len, int64_t N, int in_i16,
                          hipStream_t stream)
{
    if (hist_len < 1)
        return hipSuccess;
    const int blocks = (hist_len + 255) / 256;
    if (in_i16)
        hipLaunchKernelGGL(fir_history_kernel<true>, dim3(blocks), dim3(256), 0, stream,
                           reinterpret_cast<const f2 *>(in), reinterpret_cast<const f2 *>(hist_in),
                           reinterpret_cast<f2 *>(hist_out), hist_len, N);
    else
        hipLaunchKernelGGL(fir_history_kernel<false>, dim3(blocks), dim3(256), 0, stream,
                           reinterpret_cast<const f2 *>(in), reinterpret_cast<const f2 *>(hist_in),
                           reinterpret_cast<f2 *>(hist_out), hist_len, N);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------
// in-band power of an IQ buffer: sum of |y|^2 (what a rack-control loop would feed back into the IF
// attenuator).  HBM-bound reduction: 16-byte loads, float64 accumulation per lane (2^28 float32 squares overflow
// float32's 24 bits of exactness), wave reduction by DPP-free shuffles, one float64 atomic per workgroup.
// --------------------------------------------------------------------------------------------------------------
// Each workgroup reads whole 32 KiB chunks (8 independent 16-byte loads per thread in flight, non-temporal: the buffer is
// read once), chunks handed out grid-stride; float64 accumulation per lane, wave shuffles, one float64 atomic per workgroup.
// (Round 3: the one-load-per-iteration grid-stride loop of round 1 streamed at 5.5 TB/s; see tools/ubench_mem2.hip for why the
// shape of the loop matters.)
constexpr int POWER_UNROLL = 8;
__global__ __launch_bounds__(256) void power_kernel(const float4 *__restrict__ iq2, uint64_t pairs, const f2 *__restrict__ tail,
                                                   uint32_t tail_count, double *__restrict__ acc)
{
    double s = 0.0;
    const uint64_t chunk = 256ull * POWER_UNROLL, nchunks = pairs / chunk;
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x)
    {
        typedef float f4n __attribute__((ext_vector_type(4)));
        const f4n *p = reinterpret_cast<const f4n *>(iq2 + c * chunk + threadIdx.x);
        f4n v[POWER_UNROLL];
#pragma unroll
        for (int k = 0; k < POWER_UNROLL; k++)
            v[k] = __builtin_nontemporal_load(p + k * 256);
#pragma unroll
        for (int k = 0; k < POWER_UNROLL; k++)
            s += (double)v[k].x * (double)v[k].x + (double)v[k].y * (double)v[k].y + (double)v[k].z * (double)v[k].z +
                 (double)v[k].w * (double)v[k].w;
    }
    // the pairs behind the last whole chunk, and the odd last sample
    for (uint64_t i = nchunks * chunk + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (uint64_t)gridDim.x * 256)
    {
        const float4 v = iq2[i];
        s += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < tail_count)
    {
        const f2 v = tail[threadIdx.x];
        s += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
    }
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(acc, (part[0] + part[1]) + (part[2] + part[3]));
}

hipError_t launch_power(const void *iq, uint64_t samples, double *acc, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double), stream);
    if (e != hipSuccess || samples == 0)
        return e;
    const uint64_t pairs = samples / 2;
    uint64_t blocks = (pairs + 256 * POWER_UNROLL - 1) / (256 * POWER_UNROLL);
    if (blocks > 2048)
        blocks = 2048; // 8 workgroups per CU: enough loads in flight, few atomics
    if (blocks == 0)
        blocks = 1;
    hipLaunchKernelGGL(power_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float4 *>(iq), pairs,
                       reinterpret_cast<const f2 *>(iq) + 2 * pairs, (uint32_t)(samples & 1), acc);
    return hipGetLastError();
}

hipError_t launch_synth(void *iq, uint64_t first, uint64_t count, uint32_t channel, const float *tone10,
                        hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    ToneTable t;
    for (int i = 0; i < 10; i++)
        t.v[i] = tone10[i];
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<f2 *>(iq), first,
                       count, 0x5130303100000000ULL + (uint64_t)channel, t);
    return hipGetLastError();
}

} // namespace if_fir
