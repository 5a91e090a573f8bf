// ubench_lds.hip — prices an LDS sample read next to a stream of v_pk_fma_f32 (development tool, not product).
// Variants: FMA per read (8/16/32), read width (b128 / 2 x b64), prefetch depth (0 = use right away, 1 = one read ahead),
// waves per SIMD 1/2/4.  Also reports the real shader clock (s_memtime vs s_memrealtime) under this load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 1024;

#define FMA8(s)                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; i++)                                                                   \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(s), "s"(h));

// MODE 0: no LDS.  MODE 1: b128, no prefetch.  MODE 2: b128, one read ahead.  MODE 3: 2 x b64, one read ahead.
// NF = pk_fma per 16 bytes read / 16  (1 -> 16 fma per read i.e. R=8;  2 -> 32 i.e. R=16;  4 -> 64 i.e. R=32)
template <int MODE, int NF>
__global__ void k(f2 *out, f2 h, unsigned long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4 *l4 = reinterpret_cast<f4 *>(smem);
    for (int i = threadIdx.x; i < 17 * 256; i += blockDim.x) l4[i] = (f4){i * 1e-6f, 1.f, 2.f, 3.f};
    __syncthreads();
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (f2){threadIdx.x * 1e-9f + i, 1.f};
    const unsigned base = (unsigned)(size_t)0 + (threadIdx.x & 255) * 272;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f4 va = {1.f, 2.f, 3.f, 4.f}, vb = {1.f, 2.f, 3.f, 4.f};
    if (MODE == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(va) : "v"(base));
    if (MODE == 3) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=v"(*(f2 *)&va), "=v"(*((f2 *)&va + 1)) : "v"(base));
    for (int it = 0; it < ITERS; it++)
    {
        const unsigned ad = base + (it & 7) * 32;
        if (MODE == 0)
        {
#pragma unroll
            for (int j = 0; j < 2 * NF; j++) { FMA8(*(f2 *)&va); }
#pragma unroll
            for (int j = 0; j < 2 * NF; j++) { FMA8(*((f2 *)&va + 1)); }
        }
        else if (MODE == 1)
        {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(va) : "v"(ad));
#pragma unroll
            for (int j = 0; j < NF; j++) { FMA8(*(f2 *)&va); FMA8(*((f2 *)&va + 1)); }
            asm volatile("ds_read_b128 %0, %1 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(vb) : "v"(ad));
#pragma unroll
            for (int j = 0; j < NF; j++) { FMA8(*(f2 *)&vb); FMA8(*((f2 *)&vb + 1)); }
        }
        else if (MODE == 2)
        {
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(vb) : "v"(ad));
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(va));
#pragma unroll
            for (int j = 0; j < NF; j++) { FMA8(*(f2 *)&va); FMA8(*((f2 *)&va + 1)); }
            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(va) : "v"(ad));
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(vb));
#pragma unroll
            for (int j = 0; j < NF; j++) { FMA8(*(f2 *)&vb); FMA8(*((f2 *)&vb + 1)); }
        }
        else
        {
            asm volatile("ds_read_b64 %0, %2 offset:16\n\tds_read_b64 %1, %2 offset:24" : "=v"(*(f2 *)&vb), "=v"(*((f2 *)&vb + 1)) : "v"(ad));
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(va));
#pragma unroll
            for (int j = 0; j < NF; j++) { FMA8(*(f2 *)&va); FMA8(*((f2 *)&va + 1)); }
            asm volatile("ds_read_b64 %0, %2 offset:32\n\tds_read_b64 %1, %2 offset:40" : "=v"(*(f2 *)&va), "=v"(*((f2 *)&va + 1)) : "v"(ad));
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vb));
#pragma unroll
            for (int j = 0; j < NF; j++) { FMA8(*(f2 *)&vb); FMA8(*((f2 *)&vb + 1)); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va), "+v"(vb));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f2 s = {va.x + vb.x, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int NF>
static void run(int cus, f2 *out, unsigned long long *clk)
{
    static const char *names[] = {"nolds", "b128_now", "b128_pf1", "2xb64_pf1"};
    for (int wps : {1, 2, 4})
    {
        const int grid = cus * wps, block = 256;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const f2 h = {0.999f, 1.001f};
        hipLaunchKernelGGL((k<MODE, NF>), dim3(grid), dim3(block), 17 * 256 * 16, 0, out, h, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<MODE, NF>), dim3(grid), dim3(block), 17 * 256 * 16, 0, out, h, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        unsigned long long c[2];
        CHECK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)c[0] / (double)c[1] * 0.1; // memrealtime ticks at 100 MHz
        const double fma_per_wave = (double)ITERS * 32 * NF;
        const double tf = (double)grid * 4 * fma_per_wave * 64 * 4 / (ms * 1e-3) / 1e12;
        const double cyc = (double)c[0] / (fma_per_wave * wps); // real cycles per pk_fma per SIMD (wave 0's span)
        printf("%-10s fma/16B=%2d w/SIMD=%d  %8.4f ms  %7.2f TF  clock %.3f GHz  %.2f cyc/pk_fma\n", names[MODE], 16 * NF, wps, ms, tf, ghz, cyc);
    }
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    f2 *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, sizeof(f2) * 256 * 4 * 2048));
    CHECK(hipMalloc(&clk, 16));
    run<0, 2>(cus, out, clk);
    run<1, 1>(cus, out, clk); run<1, 2>(cus, out, clk); run<1, 4>(cus, out, clk);
    run<2, 1>(cus, out, clk); run<2, 2>(cus, out, clk); run<2, 4>(cus, out, clk);
    run<3, 1>(cus, out, clk); run<3, 2>(cus, out, clk); run<3, 4>(cus, out, clk);
    return 0;
}
