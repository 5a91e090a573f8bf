// ubench_mem.hip — achievable HBM bandwidth of the overlap-save kernel's access pattern (development tool):
// persistent waves, each reading whole 32 KB blocks as 64 rows of 512 B (8 B/lane) or 32 rows of 1 KB (16 B/lane),
// optionally writing 1/4 of the volume back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int W16, int WRITE>
__global__ __launch_bounds__(512, 2) void k_rows(const float *__restrict__ in, float *__restrict__ out, long nblocks, int waves, int run)
{
    const int lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    for (long q = gw; q * run < nblocks; q += waves)
        for (long blk = q * run; blk < (q + 1) * run && blk < nblocks; blk++)
        {
            if (W16)
            {
                const f4 *src = reinterpret_cast<const f4 *>(in + blk * 7680) + lane; // 3840 samples * 2 floats
                f4 v[32];
#pragma unroll
                for (int i = 0; i < 32; i++) v[i] = src[i * 64];
                f4 s = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 32; i++) s += v[i];
                if (WRITE)
                {
                    f4 *dst = reinterpret_cast<f4 *>(out + blk * 1920) + lane;
#pragma unroll
                    for (int i = 0; i < 7; i++) dst[i * 64] = v[i] + s;
                    if (lane < 32) dst[7 * 64] = v[7] + s;
                }
                else if (s.x == 123.456f) out[0] = s.y;
            }
            else
            {
                const f2 *src = reinterpret_cast<const f2 *>(in + blk * 7680) + lane;
                f2 v[64];
#pragma unroll
                for (int i = 0; i < 64; i++) v[i] = src[i * 64];
                f2 s = {0, 0};
#pragma unroll
                for (int i = 0; i < 64; i++) s += v[i];
                if (WRITE)
                {
                    f2 *dst = reinterpret_cast<f2 *>(out + blk * 1920) + lane;
#pragma unroll
                    for (int i = 0; i < 15; i++) dst[i * 64] = v[i] + s;
                }
                else if (s.x == 123.456f) out[0] = s.y;
            }
        }
}

// Would the HBM like its reads and writes in separate time slots?  Pure reads stream at 6.6 TB/s and pure writes at
// 6.9 TB/s, the 4:1 mix of the filter at 5.2-5.5.  Here every wave issues loads only in the "read" part of a chip-wide
// period of s_memrealtime (100 MHz) and stores only in the "write" part, so the memory sees alternating pure phases.
__global__ __launch_bounds__(512, 2) void k_rows_win(const float *__restrict__ in, float *__restrict__ out, long nblocks, int waves,
                                                    int run, unsigned period, unsigned wr_ticks)
{
    const int lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    for (long q = gw; q * run < nblocks; q += waves)
        for (long blk = q * run; blk < (q + 1) * run && blk < nblocks; blk++)
        {
            while ((unsigned)(__builtin_amdgcn_s_memrealtime() % period) >= period - wr_ticks)
                __builtin_amdgcn_s_sleep(2); // not a read slot
            const f2 *src = reinterpret_cast<const f2 *>(in + blk * 7680) + lane;
            f2 v[64];
#pragma unroll
            for (int i = 0; i < 64; i++) v[i] = src[i * 64];
            f2 s = {0, 0};
#pragma unroll
            for (int i = 0; i < 64; i++) s += v[i];
            while ((unsigned)(__builtin_amdgcn_s_memrealtime() % period) < period - wr_ticks)
                __builtin_amdgcn_s_sleep(2); // not a write slot
            f2 *dst = reinterpret_cast<f2 *>(out + blk * 1920) + lane;
#pragma unroll
            for (int i = 0; i < 15; i++) dst[i * 64] = v[i] + s;
        }
}

template <typename F> static float time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t nsamp = (size_t)1 << 28;
    float *in, *out;
    CHECK(hipMalloc(&in, nsamp * 8 + 65536));
    CHECK(hipMalloc(&out, nsamp * 2 + 65536));
    CHECK(hipMemset(in, 1, nsamp * 8 + 65536));
    const long nblocks = nsamp / 3840 - 1;
    const double rd = (double)nblocks * 32768, wr = (double)nblocks * 7680;
    for (int run : {1, 8})
        for (int wgs : {256, 512})
        {
            const int waves = wgs * 8;
            float a = time_ms([&]() { hipLaunchKernelGGL((k_rows<0, 0>), dim3(wgs), dim3(512), 0, 0, in, out, nblocks, waves, run); }, 10);
            float b = time_ms([&]() { hipLaunchKernelGGL((k_rows<1, 0>), dim3(wgs), dim3(512), 0, 0, in, out, nblocks, waves, run); }, 10);
            float c = time_ms([&]() { hipLaunchKernelGGL((k_rows<0, 1>), dim3(wgs), dim3(512), 0, 0, in, out, nblocks, waves, run); }, 10);
            float d = time_ms([&]() { hipLaunchKernelGGL((k_rows<1, 1>), dim3(wgs), dim3(512), 0, 0, in, out, nblocks, waves, run); }, 10);
            printf("run=%d wgs=%d: read-only 8B/lane %.3f ms (%.2f TB/s) | 16B/lane %.3f ms (%.2f TB/s) || read+write(1/4) 8B %.3f ms (%.2f TB/s) | 16B %.3f ms (%.2f TB/s)\n",
                   run, wgs, a, rd / a / 1e9, b, rd / b / 1e9, c, (rd + wr) / c / 1e9, d, (rd + wr) / d / 1e9);
        }
    for (int wgs : {256, 512})
        for (unsigned period : {200u, 400u, 800u, 1600u, 3200u})
            for (unsigned wr_pct : {15u, 25u, 40u})
            {
                const int waves = wgs * 8;
                const unsigned wr_ticks = period * wr_pct / 100;
                float t = time_ms([&]() { hipLaunchKernelGGL(k_rows_win, dim3(wgs), dim3(512), 0, 0, in, out, nblocks, waves, 8, period, wr_ticks); }, 5);
                printf("windowed: wgs=%d period %.1f us, write slot %u %%: read+write(1/4) 8B %.3f ms (%.2f TB/s)\n", wgs, period * 0.01,
                       wr_pct, t, (rd + wr) / t / 1e9);
            }
    return 0;
}
