// ubench_mem2.hip — what is the memory floor of the overlap-save kernel's traffic (development tool, round 2)?
// (1) plain copies in several launch shapes (why did round 1's copy fall from 5.48 to 4.7 TB/s above 1024 workgroups?)
// (2) the filter's 4:1 read:write mix as persistent waves that read whole FFT blocks and write a quarter of the volume:
//     load width 8/16 B per lane, store width 8/16 B, block = 4096 or 2048 samples, cache-policy bits, and the
//     block -> wave mapping (static interleaved, static runs, XCD-chunked), with the next block's loads issued before
//     the current block's stores as the real kernel does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t srd_t;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ srd_t make_srd(const void *p, long bytes)
{
    const unsigned long a = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const long clipped = bytes < 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffffL : bytes);
    const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)clipped);
    void *q = (void *)(((unsigned long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, n, 0x00020000);
}

// ---------------------------------------------------------------------------------------------------------------
// copies
template <int NT>
__global__ __launch_bounds__(256) void copy_oneshot(const f4 *__restrict__ in, f4 *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
    {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
        else out[i] = in[i];
    }
}
// each workgroup copies a contiguous chunk of K KiB-rows... (K float4 per thread, stride 256)
template <int K, int NT>
__global__ __launch_bounds__(256) void copy_chunk(const f4 *__restrict__ in, f4 *__restrict__ out, long n)
{
    const long base = (long)blockIdx.x * 256 * K + threadIdx.x;
    f4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = NT ? __builtin_nontemporal_load(in + base + k * 256) : in[base + k * 256];
#pragma unroll
    for (int k = 0; k < K; k++)
    {
        if (NT) __builtin_nontemporal_store(v[k], out + base + k * 256);
        else out[base + k * 256] = v[k];
    }
}
template <int K, int NT>
__global__ __launch_bounds__(256) void copy_gridstride(const f4 *__restrict__ in, f4 *__restrict__ out, long n)
{
    const long stride = (long)gridDim.x * 256 * K;
    for (long base = (long)blockIdx.x * 256 * K + threadIdx.x; base < n; base += stride)
    {
        f4 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = NT ? __builtin_nontemporal_load(in + base + k * 256) : in[base + k * 256];
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            if (NT) __builtin_nontemporal_store(v[k], out + base + k * 256);
            else out[base + k * 256] = v[k];
        }
    }
}
// the 4:1 mix in its simplest form: a workgroup reads 16 KiB and writes 4 KiB, one shot
template <int NT>
__global__ __launch_bounds__(256) void mix41_oneshot(const f4 *__restrict__ in, f4 *__restrict__ out, long nout)
{
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    const long base = (long)blockIdx.x * 1024 + threadIdx.x;
    f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = NT ? __builtin_nontemporal_load(in + base + k * 256) : in[base + k * 256];
    const f4 s = v[0] + v[1] + v[2] + v[3];
    if (NT) __builtin_nontemporal_store(s, out + o);
    else out[o] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// FFT-block shaped streaming.  NR16 = KiB per block (32: 4096 samples, 16: 2048 samples); the block starts OVLB bytes
// before its unit (overlap re-read); unit = NR16*1024 - 2048 bytes of new input, a quarter of that is written.
struct MapArgs { long nunits; int waves; int map; int run; int wgs; unsigned *queue; };
// map 6 / 7: blocks in GLOBAL order from 8 counters (one per blockIdx & 7, 256 B apart), one ticket per block, fetched one
// block ahead: consecutive blocks go to whichever waves ask next anywhere on the chip (6: singly, 7: in pairs per counter)
// FLAGS (template): 1 = the waves of a workgroup move in lockstep (s_barrier per unit), 2 = the stores of a unit go out
// ahead of the next unit's loads

__device__ __forceinline__ long unit_of(const MapArgs &m, long it, int j, int wid, int nw)
{
    const int gw = blockIdx.x * nw + wid;
    switch (m.map)
    {
    case 0: return it * m.waves + gw;                                  // static, wave-interleaved: compact chip-wide window
    case 1: return (it * m.waves + gw) * m.run + j;                    // static runs of `run` units per wave
    case 2: return it * m.waves + (long)wid * m.wgs + blockIdx.x;      // neighbours of a unit live on other workgroups
    case 5: return it == 0 ? (long)gw : m.nunits;                      // one shot: one unit per wave, the grid covers the stream
    default:
    {
        const int x = blockIdx.x & 7, wx = (blockIdx.x >> 3) * nw + wid, Wx = m.waves >> 3; // XCD-chunked window
        return it * m.waves + (long)x * Wx + wx;
    }
    }
}

template <int NR16, int LW, int SW, int LAUX, int SAUX, int WG, int MINW, int FLAGS = 0>
__global__ __launch_bounds__(WG, MINW) void k_blocks(const char *__restrict__ in, char *__restrict__ out, MapArgs m)
{
    constexpr int BLKB = NR16 * 1024, OVLB = 2048, UB = BLKB - OVLB, OB = UB / 4;
    constexpr int NV = NR16 * 2;        // dword pairs per lane... data registers: NV f2 = NR16 f4
    constexpr int NST8 = OB / 512;      // 8-byte store rows (15 or 7)
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int nw = WG / 64;
    f2 v[NV];
    f2 st[NST8 + 1];
    const int runlen = m.map == 1 ? m.run : 1;
    bool have = false;
    const bool dyn = m.map >= 6;
    unsigned *qc = m.queue + (blockIdx.x & 7) * 64;
    auto ticket_unit = [&](unsigned t) -> long {
        const long x = blockIdx.x & 7;
        return m.map == 6 ? (long)t * 8 + x : (long)(t >> 1) * 16 + 2 * x + (t & 1);
    };
    unsigned tk = 0, tk_next = 0;
    if (dyn)
    {
        if (lane == 0) tk = atomicAdd(qc, 1u);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (lane == 0) tk_next = atomicAdd(qc, 1u);   // one block ahead; read at the next iteration
    }
    long u = dyn ? ticket_unit(tk) : unit_of(m, 0, 0, wid, nw);
    long it = 0;
    int j = 0;
    auto issue_loads = [&](long uu, bool valid) { // an invalid unit reads zeros through an empty descriptor (no branch)
        const srd_t srd = make_srd(in + (valid ? uu : 0) * UB, valid ? BLKB : 0); // (the buffer starts OVLB before unit 0: see main)
        if (LW == 16)
        {
#pragma unroll
            for (int r = 0; r < NR16; r++)
            {
                const u4 w = __builtin_amdgcn_raw_buffer_load_b128(srd, lane * 16u, r * 1024, LAUX);
                v[2 * r] = (f2){__uint_as_float(w[0]), __uint_as_float(w[1])};
                v[2 * r + 1] = (f2){__uint_as_float(w[2]), __uint_as_float(w[3])};
            }
        }
        else
        {
#pragma unroll
            for (int r = 0; r < NV; r++)
            {
                const u2 w = __builtin_amdgcn_raw_buffer_load_b64(srd, lane * 8u, r * 512, LAUX);
                v[r] = (f2){__uint_as_float(w[0]), __uint_as_float(w[1])};
            }
        }
    };
    have = u < m.nunits;
    issue_loads(u, have);
    while (have)
    {
        // consume the block (stands for the transform): a reduction that depends on every row
        f2 s = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NV; r++) s += v[r];
#pragma unroll
        for (int r = 0; r < NST8 + 1; r++) st[r] = v[r] + s;
        __builtin_amdgcn_sched_barrier(0);
        // next unit of this wave
        long un;
        j++;
        if (j >= runlen) { j = 0; it++; }
        un = unit_of(m, it, j, wid, nw);
        if (dyn)
        {
            un = ticket_unit(__builtin_amdgcn_readfirstlane(tk_next));
            if (lane == 0) tk_next = atomicAdd(qc, 1u);
        }
        const bool more = un < m.nunits;
        if (FLAGS & 1)
            __builtin_amdgcn_s_barrier();
        if (!(FLAGS & 2))
            issue_loads(un, more);
        __builtin_amdgcn_sched_barrier(0);
        // stores of the current unit behind the next unit's loads (flags & 2: ahead of them)
        const srd_t osrd = make_srd(out + u * OB, OB);
        if (SW == 16)
        {
#pragma unroll
            for (int r = 0; r < NST8 / 2; r++)
            {
                u4 w = {__float_as_uint(st[2 * r].x), __float_as_uint(st[2 * r].y), __float_as_uint(st[2 * r + 1].x), __float_as_uint(st[2 * r + 1].y)};
                __builtin_amdgcn_raw_buffer_store_b128(w, osrd, lane * 16u, r * 1024, SAUX);
            }
            if (NST8 & 1) // the odd half row: lanes 32..63 fall outside the descriptor and are dropped
            {
                u4 w = {__float_as_uint(st[NST8 - 1].x), __float_as_uint(st[NST8 - 1].y), __float_as_uint(st[NST8].x), __float_as_uint(st[NST8].y)};
                __builtin_amdgcn_raw_buffer_store_b128(w, osrd, lane * 16u, (NST8 / 2) * 1024, SAUX);
            }
        }
        else
        {
#pragma unroll
            for (int r = 0; r < NST8; r++)
            {
                u2 w = {__float_as_uint(st[r].x), __float_as_uint(st[r].y)};
                __builtin_amdgcn_raw_buffer_store_b64(w, osrd, lane * 8u, r * 512, SAUX);
            }
        }
        if (FLAGS & 2)
        {
            __builtin_amdgcn_sched_barrier(0);
            issue_loads(un, more);
        }
        u = un;
        have = more;
    }
}

template <typename F> static float time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); launch(); CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++)
    {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    CHECK(hipGetLastError());
    std::sort(t.begin(), t.end());
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

static char *g_in, *g_out;
static unsigned *g_queue;
static const size_t NSAMP = (size_t)1 << 28;

template <int NR16, int LW, int SW, int LAUX, int SAUX, int WG, int MINW, int FLAGS = 0>
static void run_blocks(const char *tag, int wgs, int map, int run)
{
    constexpr long UB = NR16 * 1024 - 2048;
    MapArgs m;
    m.nunits = (long)(NSAMP * 8 / UB);
    m.waves = wgs * (WG / 64);
    m.map = map; m.run = run; m.wgs = wgs; m.queue = g_queue;
    auto kern = k_blocks<NR16, LW, SW, LAUX, SAUX, WG, MINW, FLAGS>;
    const float t = time_ms([&]() { if (map >= 6) CHECK(hipMemsetAsync(g_queue, 0, 8 * 256, 0)); hipLaunchKernelGGL(kern, dim3(wgs), dim3(WG), 0, 0, g_in, g_out, m); }, 9);
    const double alg = (double)m.nunits * UB * 1.25;
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
    printf("blocks %-10s pts=%d LW=%2d SW=%2d laux=%2d saux=%2d WG=%d wgs=%4d map=%d run=%d vgpr=%3d : %.3f ms  %.2f TB/s alg (frac %.3f)\n", tag,
           NR16 * 128, LW, SW, LAUX, SAUX, WG, wgs, map, run, fa.numRegs, t, alg / t / 1e9, alg / t / 1e9 / 8.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const size_t inb = NSAMP * 8, outb = NSAMP * 8; // the copy tests write a full-size buffer
    CHECK(hipMalloc(&g_in, inb + (1 << 20)));
    CHECK(hipMalloc(&g_out, outb + (1 << 20)));
    CHECK(hipMemset(g_in, 1, inb + (1 << 20)));
    CHECK(hipMemset(g_out, 0, outb + (1 << 20)));
    g_in += 4096; // blocks start OVLB before their unit
    CHECK(hipMalloc(&g_queue, 8 * 256));
    const bool all = argc < 2;
    const char *sel = all ? "" : argv[1];
    auto want = [&](const char *s) { return all || strstr(sel, s); };
    if (want("copy"))
    {
        const long n16 = (long)(inb / 16);
        const f4 *in = (const f4 *)g_in; f4 *out = (f4 *)g_out;
        float t;
        t = time_ms([&]() { CHECK(hipMemcpyAsync(g_out, g_in, inb, hipMemcpyDeviceToDevice, 0)); }, 9);
        printf("copy hipMemcpyDtoD 2GiB           : %.3f ms %.2f TB/s (r+w)\n", t, 2.0 * inb / t / 1e9);
        t = time_ms([&]() { CHECK(hipMemsetAsync(g_out, 0, inb, 0)); }, 9);
        printf("fill hipMemset 2GiB               : %.3f ms %.2f TB/s (w)\n", t, 1.0 * inb / t / 1e9);
        t = time_ms([&]() { hipLaunchKernelGGL(copy_oneshot<0>, dim3(n16 / 256), dim3(256), 0, 0, in, out, n16); }, 9);
        printf("copy oneshot 1xf4/thread          : %.3f ms %.2f TB/s (r+w)\n", t, 2.0 * inb / t / 1e9);
        t = time_ms([&]() { hipLaunchKernelGGL(copy_oneshot<1>, dim3(n16 / 256), dim3(256), 0, 0, in, out, n16); }, 9);
        printf("copy oneshot 1xf4/thread nt       : %.3f ms %.2f TB/s (r+w)\n", t, 2.0 * inb / t / 1e9);
        t = time_ms([&]() { hipLaunchKernelGGL((copy_chunk<4, 0>), dim3(n16 / 1024), dim3(256), 0, 0, in, out, n16); }, 9);
        printf("copy oneshot 4xf4/thread          : %.3f ms %.2f TB/s (r+w)\n", t, 2.0 * inb / t / 1e9);
        t = time_ms([&]() { hipLaunchKernelGGL((copy_chunk<4, 1>), dim3(n16 / 1024), dim3(256), 0, 0, in, out, n16); }, 9);
        printf("copy oneshot 4xf4/thread nt       : %.3f ms %.2f TB/s (r+w)\n", t, 2.0 * inb / t / 1e9);
        t = time_ms([&]() { hipLaunchKernelGGL((copy_chunk<8, 0>), dim3(n16 / 2048), dim3(256), 0, 0, in, out, n16); }, 9);
        printf("copy oneshot 8xf4/thread          : %.3f ms %.2f TB/s (r+w)\n", t, 2.0 * inb / t / 1e9);
        for (int g : {256, 512, 1024, 2048, 4096, 8192})
        {
            t = time_ms([&]() { hipLaunchKernelGGL((copy_gridstride<4, 0>), dim3(g), dim3(256), 0, 0, in, out, n16); }, 9);
            float t2 = time_ms([&]() { hipLaunchKernelGGL((copy_gridstride<4, 1>), dim3(g), dim3(256), 0, 0, in, out, n16); }, 9);
            float t3 = time_ms([&]() { hipLaunchKernelGGL((copy_gridstride<1, 0>), dim3(g), dim3(256), 0, 0, in, out, n16); }, 9);
            printf("copy gridstride grid=%5d: 4xf4 %.3f ms %.2f TB/s | 4xf4 nt %.3f ms %.2f TB/s | 1xf4 %.3f ms %.2f TB/s\n", g, t,
                   2.0 * inb / t / 1e9, t2, 2.0 * inb / t2 / 1e9, t3, 2.0 * inb / t3 / 1e9);
        }
        // 512 MiB copies (the output size of the filter)
        const long n16s = n16 / 4;
        t = time_ms([&]() { hipLaunchKernelGGL((copy_chunk<4, 0>), dim3(n16s / 1024), dim3(256), 0, 0, in, out, n16s); }, 9);
        printf("copy oneshot 4xf4/thread 512MiB   : %.3f ms %.2f TB/s (r+w)\n", t, 2.0 * (inb / 4) / t / 1e9);
        const long nout = n16 / 4;
        t = time_ms([&]() { hipLaunchKernelGGL(mix41_oneshot<0>, dim3(nout / 256), dim3(256), 0, 0, in, out, nout); }, 9);
        printf("mix 4:1 oneshot (16K in, 4K out per WG)    : %.3f ms %.2f TB/s (frac %.3f)\n", t, 1.25 * inb / t / 1e9, 1.25 * inb / t / 8e12 * 1e3);
        t = time_ms([&]() { hipLaunchKernelGGL(mix41_oneshot<1>, dim3(nout / 256), dim3(256), 0, 0, in, out, nout); }, 9);
        printf("mix 4:1 oneshot nt                         : %.3f ms %.2f TB/s (frac %.3f)\n", t, 1.25 * inb / t / 1e9, 1.25 * inb / t / 8e12 * 1e3);
        fflush(stdout);
    }
    if (want("oneshot"))
    {
        // the filter's blocks handed out by the hardware dispatcher instead of a persistent grid: one 4096-point block per
        // wave, 8 waves per workgroup, nunits / 8 workgroups in dispatch order (no tables to reload here; the real
        // kernel would pay an 82 KB table load per workgroup)
        const int wgs = (int)((NSAMP * 8 / (32 * 1024 - 2048)) / 8);
        run_blocks<32, 8, 8, 0, 0, 512, 2>("4k 1shot", wgs, 5, 1);
        run_blocks<32, 16, 16, 0, 0, 512, 2>("4k 1shot", wgs, 5, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2>("4k 1shot ntLS", wgs, 5, 1);
        run_blocks<32, 8, 8, 0, 0, 512, 2>("4k persist", 256, 0, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2>("4k persist ntLS", 256, 0, 1);
        run_blocks<16, 16, 16, 2, 2, 512, 4>("2k 1shot ntLS", (int)((NSAMP * 8 / (16 * 1024 - 2048)) / 8), 5, 1);
    }
    if (!all && strstr(sel, "soak"))
    {
        // soak <seconds> [mix]: one pattern back to back for a while (power sampling from outside, tools/power_attr.sh):
        // the filter's persistent skeleton (16-byte rows, nt loads and stores, 256 workgroups) or the one-shot 4:1 mix
        const double secs = argc > 2 ? atof(argv[2]) : 8.0;
        const bool mix = argc > 3 && strstr(argv[3], "mix");
        constexpr long UB = 32 * 1024 - 2048;
        MapArgs m;
        m.nunits = (long)(NSAMP * 8 / UB);
        m.waves = 256 * 8; m.map = 0; m.run = 1; m.wgs = 256; m.queue = g_queue;
        const long nout = (long)(inb / 16) / 4;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const auto t0 = std::chrono::steady_clock::now();
        float ms = 0;
        long launches = 0;
        do
        {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 200; i++)
            {
                if (mix)
                    hipLaunchKernelGGL(mix41_oneshot<1>, dim3(nout / 256), dim3(256), 0, 0, (const f4 *)g_in, (f4 *)g_out, nout);
                else
                    hipLaunchKernelGGL((k_blocks<32, 16, 16, 2, 2, 512, 2, 0>), dim3(256), dim3(512), 0, 0, g_in, g_out, m);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            launches += 200;
        } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs);
        printf("soak %s: %.4f ms per launch (last 200 of %ld), %.2f TB/s algorithmic\n", mix ? "mix 4:1 one-shot nt" : "persistent skeleton ntLS", ms / 200,
               launches, 1.25 * inb / (ms / 200) / 1e9);
        return 0;
    }
    if (want("shape"))
    {
        // what makes the dispatcher-ordered launch faster than the persistent grid?  lockstep workgroups / stores first
        const int wgs1 = (int)((NSAMP * 8 / (32 * 1024 - 2048)) / 8);
        run_blocks<32, 16, 16, 2, 2, 512, 2>("1shot", wgs1, 5, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2>("persist", 256, 0, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2, 1>("persist lockstep", 256, 0, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2, 2>("persist st-first", 256, 0, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2, 3>("persist lock+stf", 256, 0, 1);
        run_blocks<32, 8, 8, 2, 2, 512, 2>("persist 8B", 256, 0, 1);
        run_blocks<32, 8, 8, 2, 2, 512, 2, 1>("persist 8B lock", 256, 0, 1);
        run_blocks<32, 8, 8, 2, 2, 512, 2, 2>("persist 8B stf", 256, 0, 1);
        run_blocks<32, 8, 8, 2, 2, 512, 2>("1shot 8B", wgs1, 5, 1);
        // blocks in global order from sharded per-block tickets (8 counters)
        run_blocks<32, 8, 8, 2, 2, 512, 2>("persist 8B gq1", 256, 6, 1);
        run_blocks<32, 8, 8, 2, 2, 512, 2>("persist 8B gq2", 256, 7, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2>("persist gq1", 256, 6, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2>("persist gq2", 256, 7, 1);
        run_blocks<32, 8, 8, 2, 2, 512, 2>("persist 8B", 256, 0, 1);
    }
    if (want("blocks"))
    {
        // baseline of round 1's kernel: 4096-point blocks, 8-byte rows, 256 workgroups of 8 waves, runs of 8
        for (int map : {1, 0, 2, 3})
        {
            const int run = map == 1 ? 8 : 1;
            run_blocks<32, 8, 8, 0, 0, 512, 2>("4k", 256, map, run);
            run_blocks<32, 16, 8, 0, 0, 512, 2>("4k", 256, map, run);
            run_blocks<32, 16, 16, 0, 0, 512, 2>("4k", 256, map, run);
            run_blocks<32, 8, 16, 0, 0, 512, 2>("4k", 256, map, run);
        }
        // 2048-point blocks at 4 waves per SIMD (two 512-thread workgroups per CU)
        for (int map : {1, 0, 3})
        {
            const int run = map == 1 ? 8 : 1;
            run_blocks<16, 8, 8, 0, 0, 512, 4>("2k", 512, map, run);
            run_blocks<16, 16, 8, 0, 0, 512, 4>("2k", 512, map, run);
            run_blocks<16, 16, 16, 0, 0, 512, 4>("2k", 512, map, run);
            run_blocks<16, 16, 16, 0, 0, 512, 4>("2k", 256, map, run);
        }
        // 1024-thread workgroups (16 waves, one workgroup per CU) for the 2048-point shape
        run_blocks<16, 16, 16, 0, 0, 1024, 4>("2k", 256, 0, 1);
        run_blocks<16, 16, 16, 0, 0, 1024, 4>("2k", 256, 3, 1);
        // cache-policy bits on the best-guess shape (aux: 1 = sc0, 2 = nt, 16 = sc1)
        run_blocks<32, 16, 16, 2, 0, 512, 2>("4k ntL", 256, 0, 1);
        run_blocks<32, 16, 16, 0, 2, 512, 2>("4k ntS", 256, 0, 1);
        run_blocks<32, 16, 16, 2, 2, 512, 2>("4k ntLS", 256, 0, 1);
        run_blocks<32, 16, 16, 0, 16, 512, 2>("4k sc1S", 256, 0, 1);
        run_blocks<32, 16, 16, 0, 17, 512, 2>("4k sc01S", 256, 0, 1);
        run_blocks<32, 16, 16, 16, 0, 512, 2>("4k sc1L", 256, 0, 1);
        run_blocks<16, 16, 16, 2, 0, 512, 4>("2k ntL", 512, 0, 1);
        run_blocks<16, 16, 16, 0, 2, 512, 4>("2k ntS", 512, 0, 1);
        run_blocks<16, 16, 16, 2, 2, 512, 4>("2k ntLS", 512, 0, 1);
        run_blocks<16, 16, 16, 0, 16, 512, 4>("2k sc1S", 512, 0, 1);
        run_blocks<16, 16, 16, 0, 17, 512, 4>("2k sc01S", 512, 0, 1);
    }
    return 0;
}
