// ubench_dep.hip — how much instruction-level parallelism does a v_pk_fma_f32 stream need on MI355X?  (development tool; round 5)
// K independent dependent chains of v_pk_fma_f32 per wave (K = 1, 2, 3, 4, 6, 8, 16), at 1, 2 and 4 waves per SIMD: wave-cycles per
// instruction and the SIMD's issue rate.  One workgroup per CU (64 KB of LDS requested), W waves per SIMD = 4 W waves per workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_dep.hip -o tools/ubench_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 4096, PER = 48; // instructions per iteration (a multiple of every K)

template <int K>
__global__ void k_dep(f2 *out, f2 h, unsigned long long *cyc)
{
    extern __shared__ char smem[];
    f2 a[K];
#pragma unroll
    for (int i = 0; i < K; i++) a[i] = (f2){threadIdx.x * 1e-9f + i, 1.f};
    f2 x = {1.0f + threadIdx.x * 1e-9f, 0.999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int rep = 0; rep < PER / K; rep++)
#pragma unroll
            for (int i = 0; i < K; i++)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(h)); // a = a * x + h: each chain depends on itself
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < K; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (smem[0] == 77) out[0] = s; // (keeps the LDS request)
}

template <int K>
void run(int wps, f2 *out, unsigned long long *cyc)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dep<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int w = 0; w < 2; w++)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_dep<K>, dim3(256), dim3(256 * wps), 100 * 1024, 0, out, (f2){1e-6f, 1e-6f}, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double instr = (double)ITERS * PER;
    printf("K=%2d chains, %d wave(s)/SIMD: %.2f shader cycles per instruction and wave -> %.2f cycles per instruction and SIMD; %.1f TFLOP/s\n", K, wps,
           (double)c / instr, (double)c / instr / wps, 256.0 * 4 * wps * instr * 64 * 4 / (ms * 1e-3) / 1e12);
}

int main()
{
    f2 *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * sizeof(f2))); CHECK(hipMalloc(&cyc, 8));
    for (int wps : {1, 2, 4})
    {
        run<1>(wps, out, cyc); run<2>(wps, out, cyc); run<3>(wps, out, cyc); run<4>(wps, out, cyc);
        run<6>(wps, out, cyc); run<8>(wps, out, cyc); run<16>(wps, out, cyc);
    }
    return 0;
}
