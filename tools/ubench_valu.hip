// ubench_valu.hip — MI355X FP32 VALU issue-rate microbenchmark (development tool, not product).
// Measures v_fma_f32 / v_pk_fma_f32 (VGPR and SGPR-tap forms) and a pk_fma + ds_read_b128 mix at 1/2/4/8 waves
// per SIMD, to size the direct-form FIR kernel.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;

// 16 independent v_fma_f32 per iteration x 4 = 64 instr / iter
__global__ void k_fma(float *out, float h0, float h1)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-9f + i;
    float x = threadIdx.x * 1e-3f;
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(h0));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + h1;
}

// 16 independent v_pk_fma_f32 (all VGPR) x 4 = 64 instr / iter
__global__ void k_pkfma_v(f2 *out, float h0)
{
    f2 a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (f2){threadIdx.x * 1e-9f + i, 1.f};
    f2 x = {threadIdx.x * 1e-3f, 0.5f};
    f2 h = {h0, h0};
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(h));
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same with the tap in an SGPR pair, broadcast with op_sel_hi (what the FIR kernel issues)
__global__ void k_pkfma_s(f2 *out, f2 h)
{
    f2 a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (f2){threadIdx.x * 1e-9f + i, 1.f};
    f2 x = {threadIdx.x * 1e-3f, 0.5f};
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "s"(h));
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 8 accumulators: per iteration 4 x (1 ds_read_b128 + 16 pk_fma using the two samples read), like the FIR walk
__global__ void k_mix(f2 *out, f2 h)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4 *l4 = reinterpret_cast<f4 *>(smem);
    for (int i = threadIdx.x; i < 17 * 256; i += blockDim.x) l4[i] = (f4){i * 1e-6f, 1.f, 2.f, 3.f};
    __syncthreads();
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (f2){threadIdx.x * 1e-9f + i, 1.f};
    const char *base = smem + (threadIdx.x & 255) * 272;
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
        {
            const f4 v = *reinterpret_cast<const f4 *>(base + rep * 16 + (it & 7) * 32);
            const f2 s0 = {v.x, v.y}, s1 = {v.z, v.w};
#pragma unroll
            for (int i = 0; i < 8; i++)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(s0), "s"(h));
#pragma unroll
            for (int i = 0; i < 8; i++)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(s1), "s"(h));
        }
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// streaming copy for the achievable-HBM number (float4 per lane, grid-stride)
__global__ void k_copy(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_read(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n)
{
    f4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s.x == 123.456f) out[0] = s;
}

template <typename F>
static float time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

// 16 independent v_pk_add_f32 x 4 per iteration (the butterfly instruction of the FFT kernel)
__global__ void k_pkadd(f2 *out, float h0)
{
    f2 a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (f2){threadIdx.x * 1e-9f + i, 1.f};
    f2 x = {threadIdx.x * 1e-3f + h0, 0.5f};
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 32 v_permlane32_swap + 32 v_permlane16_swap per iteration on 16 independent register pairs (exchange 1 of the FFT)
__global__ void k_permlane(f2 *out, float h0)
{
    unsigned a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        a[i] = threadIdx.x * 3 + i;
        b[i] = threadIdx.x * 5 + i + (unsigned)h0;
    }
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int rep = 0; rep < 2; rep++)
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (f2){(float)s, 0.f};
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s %s CUs=%d clock=%d MHz\n", p.name, p.gcnArchName, cus, p.clockRate / 1000);
    f2 *out;
    CHECK(hipMalloc(&out, sizeof(f2) * 256 * 8 * 2048));
    const f2 h = {0.999f, 1.001f};
    printf("%-10s %6s %10s %10s %10s\n", "kernel", "w/SIMD", "ms", "TFLOP/s", "cyc/instr@2.4GHz");
    for (int wps : {1, 2, 4, 8})
    {
        const int block = 256; // 4 waves -> one per SIMD; blocks/CU = wps
        const int grid = cus * wps;
        const double instr = (double)ITERS * 64; // per wave
        struct { const char *name; double flop_per_instr_lane; int which; } ks[] = {{"fma", 2, 0}, {"pkfma_v", 4, 1}, {"pkfma_s", 4, 2}, {"mix", 4, 3}, {"pkadd", 2, 4}, {"permlane", 0, 5}};
        for (auto &k : ks)
        {
            float ms = time_ms([&]() {
                if (k.which == 0) hipLaunchKernelGGL(k_fma, dim3(grid), dim3(block), 0, 0, (float *)out, 0.999f, 0.f);
                if (k.which == 1) hipLaunchKernelGGL(k_pkfma_v, dim3(grid), dim3(block), 0, 0, out, 0.999f);
                if (k.which == 2) hipLaunchKernelGGL(k_pkfma_s, dim3(grid), dim3(block), 0, 0, out, h);
                if (k.which == 3) hipLaunchKernelGGL(k_mix, dim3(grid), dim3(block), 17 * 256 * 16, 0, out, h);
                if (k.which == 4) hipLaunchKernelGGL(k_pkadd, dim3(grid), dim3(block), 0, 0, out, 0.999f);
                if (k.which == 5) hipLaunchKernelGGL(k_permlane, dim3(grid), dim3(block), 0, 0, out, 0.999f);
            }, 5);
            const double waves = (double)grid * block / 64;
            const double flops = waves * instr * 64 * k.flop_per_instr_lane;
            // cycles per instruction per SIMD: time * clock / (instr per wave * waves per SIMD)
            const double cyc = ms * 1e-3 * 2.4e9 / (instr * wps);
            printf("%-10s %6d %10.4f %10.2f %10.2f\n", k.name, wps, ms, flops / (ms * 1e-3) / 1e12, cyc);
        }
    }
    // HBM copy / read
    const size_t bytes = (size_t)2 << 30;
    f4 *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 0, bytes));
    for (int g : {cus * 4, cus * 8, cus * 16, cus * 32})
    {
        float ms = time_ms([&]() { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, bytes / 16); }, 5);
        float mr = time_ms([&]() { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, b, bytes / 16); }, 5);
        printf("copy grid=%6d: %.3f ms  %.2f TB/s (r+w) | read-only %.3f ms %.2f TB/s\n", g, ms, 2.0 * bytes / (ms * 1e-3) / 1e12, mr, 1.0 * bytes / (mr * 1e-3) / 1e12);
    }
    return 0;
}
