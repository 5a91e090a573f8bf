// ubench_energy.hip — energy per instruction class on MI355X, and of the two ways to do "16-point transform + 15 twiddles"
// (development tool, not product; DESIGN §3.4 finding 12).  Every variant runs the overlap-save kernel's occupancy (one
// 512-thread workgroup per CU, 2 waves per SIMD, 64 complex values per lane in registers, twiddles from LDS) on random data
// for a given number of seconds; tools/energy_attr.sh samples rocm-smi beside it: energy per group = (W - idle W) / (groups/s).
//   add | mul | fma     64 groups of 110 independent v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 per iteration
//   old                 fft16 (radix-4 x radix-4: 64 pk_add + 8 constant twiddles) + 15 twiddle multiplies (c, s) from LDS:
//                       110 packed instructions and 15 LDS reads per group -- what passes 1 and 2 of the kernel execute
//   new                 fft16 with the twiddles on the INPUTS of its butterflies in (cos, tan) form: every butterfly output is a
//                       chain of fused multiply-adds (88 packed instructions, 15 LDS reads per group)
//   scale               only the re-normalisation both transform variants carry (16 pk_mul per group): subtract it
// Build: hipcc --offload-arch=gfx950 -O3.   Usage: ubench_energy <variant> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float cf __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- the kernel's arithmetic helpers (if_fir_fft.hip) ---------------------------------------------------------------
template <bool CONJ>
__device__ __forceinline__ cf cmul_v(cf a, cf w)
{
    cf t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    if (CONJ)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
template <bool CONJ>
__device__ __forceinline__ cf cmul_s(cf a, cf w)
{
    cf t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    if (CONJ)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}
template <bool INV>
__device__ __forceinline__ cf add_rot(cf a, cf b)
{
    cf d;
    if (INV)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <bool INV>
__device__ __forceinline__ cf sub_rot(cf a, cf b) { return add_rot<!INV>(a, b); }
template <bool INV>
__device__ __forceinline__ void bfly4(cf a, cf b, cf c, cf d, cf &u0, cf &u1, cf &u2, cf &u3)
{
    const cf t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    u0 = t0 + t2;
    u1 = add_rot<INV>(t1, t3);
    u2 = t0 - t2;
    u3 = sub_rot<INV>(t1, t3);
}
template <bool INV>
__device__ __forceinline__ void bfly4_crot(cf a, cf b, cf c, cf d, cf &u0, cf &u1, cf &u2, cf &u3)
{
    const cf t0 = add_rot<INV>(a, c), t1 = sub_rot<INV>(a, c), t2 = b + d, t3 = b - d;
    u0 = t0 + t2;
    u1 = add_rot<INV>(t1, t3);
    u2 = t0 - t2;
    u3 = sub_rot<INV>(t1, t3);
}
template <bool INV>
__device__ __forceinline__ void fft16(cf (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    cf y[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        bfly4<INV>(v[i], v[i + 4], v[i + 8], v[i + 12], y[0][i], y[1][i], y[2][i], y[3][i]);
    y[1][1] = cmul_s<INV>(y[1][1], (cf){C1, -S1});
    y[2][1] = cmul_s<INV>(y[2][1], (cf){R, -R});
    y[3][1] = cmul_s<INV>(y[3][1], (cf){S1, -C1});
    y[1][2] = cmul_s<INV>(y[1][2], (cf){R, -R});
    y[3][2] = cmul_s<INV>(y[3][2], (cf){-R, -R});
    y[1][3] = cmul_s<INV>(y[1][3], (cf){S1, -C1});
    y[2][3] = cmul_s<INV>(y[2][3], (cf){-R, -R});
    y[3][3] = cmul_s<INV>(y[3][3], (cf){-C1, S1});
    bfly4<INV>(y[0][0], y[0][1], y[0][2], y[0][3], v[0], v[4], v[8], v[12]);
    bfly4<INV>(y[1][0], y[1][1], y[1][2], y[1][3], v[1], v[5], v[9], v[13]);
    bfly4_crot<INV>(y[2][0], y[2][1], y[2][2], y[2][3], v[2], v[6], v[10], v[14]);
    bfly4<INV>(y[3][0], y[3][1], y[3][2], y[3][3], v[3], v[7], v[11], v[15]);
}

// ---- the (cos, tan) form --------------------------------------------------------------------------------------------
// A twiddle w = c (1 + j t) is kept as E = (c, t).  u = x + j t x is one packed FMA, a +- c u another: a radix-4 butterfly whose
// inputs 1..3 carry twiddles w1, w2, w3 costs 11 packed FMAs (the third entry holds (c3 / c1, t3)):
//   u2 = x2 (1 + j t2);  t0 = x0 + c2 u2;  t1 = x0 - c2 u2;  u1, u3 likewise;  v+- = u1 +- (c3/c1) u3;
//   X0 = t0 + c1 v+;  X2 = t0 - c1 v+;  X1 = t1 -+ j c1 v-;  X3 = t1 +- j c1 v-
template <bool INV>
__device__ __forceinline__ cf tw_u(cf x, cf e) // x (1 + j t) (forward) / x (1 - j t) (inverse: conjugate twiddles)
{
    cf d;
    if (INV)
        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(e));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(x), "v"(e));
    return d;
}
template <bool NEG>
__device__ __forceinline__ cf tw_ac(cf a, cf u, cf e) // a +- e.x u
{
    cf d;
    if (NEG)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(u), "v"(e), "v"(a));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(u), "v"(e), "v"(a));
    return d;
}
template <bool PLUSJ>
__device__ __forceinline__ cf tw_ajc(cf a, cf u, cf e) // a +- j e.x u
{
    cf d;
    if (PLUSJ) // (a.x - c u.y, a.y + c u.x)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(u), "v"(e), "v"(a));
    else       // (a.x + c u.y, a.y - c u.x)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(u), "v"(e), "v"(a));
    return d;
}
template <bool INV>
__device__ __forceinline__ void bfly4_tw(cf x0, cf x1, cf x2, cf x3, cf e1, cf e2, cf e3, cf &X0, cf &X1, cf &X2, cf &X3)
{
    const cf u2 = tw_u<INV>(x2, e2);
    const cf t0 = tw_ac<false>(x0, u2, e2), t1 = tw_ac<true>(x0, u2, e2);
    const cf u1 = tw_u<INV>(x1, e1), u3 = tw_u<INV>(x3, e3);
    const cf vp = tw_ac<false>(u1, u3, e3), vm = tw_ac<true>(u1, u3, e3); // e3.x = c3 / c1
    X0 = tw_ac<false>(t0, vp, e1);
    X2 = tw_ac<true>(t0, vp, e1);
    X1 = tw_ajc<INV>(t1, vm, e1);  // forward: t1 - j c1 v-
    X3 = tw_ajc<!INV>(t1, vm, e1);
}
// 16-point transform of v[j] b^j (b = the lane's base twiddle): e[0..2] = b^4, b^8, b^12; e[3 + 3 q + (i - 1)] = b^i W16^(i q)
template <bool INV>
__device__ __forceinline__ void fft16_tw(cf (&v)[16], const cf *e, int stride)
{
    cf y[4][4];
    {
        const cf e1 = e[0], e2 = e[stride], e3 = e[2 * stride];
#pragma unroll
        for (int i = 0; i < 4; i++)
            bfly4_tw<INV>(v[i], v[i + 4], v[i + 8], v[i + 12], e1, e2, e3, y[0][i], y[1][i], y[2][i], y[3][i]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        bfly4_tw<INV>(y[q][0], y[q][1], y[q][2], y[q][3], e[(3 + 3 * q) * stride], e[(4 + 3 * q) * stride], e[(5 + 3 * q) * stride],
                      v[q], v[q + 4], v[q + 8], v[q + 12]);
}

constexpr int LDS_BYTES = 150 * 1024;
enum { V_ADD, V_MUL, V_FMA, V_OLD, V_NEW, V_SCALE, V_CHECK };

template <int VAR>
__global__ __launch_bounds__(512, 2) void k_energy(const cf *__restrict__ tab, const cf *__restrict__ in, cf *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *lt = reinterpret_cast<cf *>(smem);
    for (int i = threadIdx.x; i < 4 * 15 * 64; i += 512)
        lt[i] = tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    cf r[64];
#pragma unroll
    for (int i = 0; i < 64; i++)
        r[i] = in[(size_t)i * 512 + threadIdx.x];
    const cf sc = {0.25f, 0.25f};
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int g = 0; g < 4; g++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
                t[j] = r[4 * j + g];
            if constexpr (VAR == V_OLD)
            {
                fft16<false>(t);
#pragma unroll
                for (int j = 1; j < 16; j++)
                    t[j] = cmul_v<false>(t[j], lt[(g * 15 + j - 1) * 64 + lane]);
            }
            else if constexpr (VAR == V_NEW)
            {
                fft16_tw<false>(t, lt + g * 15 * 64 + lane, 64);
            }
            else if constexpr (VAR == V_ADD || VAR == V_MUL || VAR == V_FMA)
            {
                // 110 independent packed instructions: sources t[0..7] (never written: the values stay the random inputs),
                // destinations t[8..15] ("+v": kept in eight distinct registers)
#pragma unroll
                for (int k = 0; k < 110; k++)
                {
                    const int a = 8 + (k & 7), b = (k + 3) & 7, c = (k + 5) & 7, d = (k + 6) & 7;
                    if (VAR == V_ADD)
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(t[a]) : "v"(t[b]), "v"(t[c]));
                    else if (VAR == V_MUL)
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(t[a]) : "v"(t[b]), "v"(t[c]));
                    else
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(t[a]) : "v"(t[b]), "v"(t[c]), "v"(t[d]));
                }
            }
            if constexpr (VAR == V_OLD || VAR == V_NEW || VAR == V_SCALE)
            {
#pragma unroll
                for (int j = 0; j < 16; j++)
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t[j]) : "v"(sc));
            }
#pragma unroll
            for (int j = 0; j < 16; j++)
                r[4 * j + g] = t[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 64; i++)
        out[((size_t)blockIdx.x * 64 + i) * 512 + threadIdx.x] = r[i];
}

// one application of both transform forms on the same data: the outputs must agree (the twiddle tables describe the same b)
__global__ __launch_bounds__(512, 2) void k_check(const cf *__restrict__ tab_old, const cf *__restrict__ tab_new, const cf *__restrict__ in,
                                                  cf *__restrict__ out_old, cf *__restrict__ out_new)
{
    const int lane = threadIdx.x & 63;
    cf a[16], b[16];
#pragma unroll
    for (int j = 0; j < 16; j++)
        a[j] = b[j] = in[(size_t)j * 512 + threadIdx.x];
    // old: twiddles on the OUTPUTS of the previous transform = on the inputs of this one: multiply first, then transform
#pragma unroll
    for (int j = 1; j < 16; j++)
        a[j] = cmul_v<false>(a[j], tab_old[(j - 1) * 64 + lane]);
    fft16<false>(a);
    fft16_tw<false>(b, tab_new + lane, 64);
#pragma unroll
    for (int j = 0; j < 16; j++)
    {
        out_old[(size_t)j * 512 + threadIdx.x] = a[j];
        out_new[(size_t)j * 512 + threadIdx.x] = b[j];
    }
}

static void tan_entry(double th, float *e, double c_ref, bool ratio)
{
    // (c, t) with c = cos(th), t = tan(th); an exact zero of the cosine is replaced by 2^-30 (its own contribution is below
    // rounding, the tangent stays finite); ratio: the first component is c / c_ref instead (third input of a butterfly)
    double c = cos(th), s = sin(th);
    if (fabs(c) < 9.3e-10)
        c = (c < 0 ? -1.0 : 1.0) * 9.313225746154785e-10;
    const float cf_ = (float)c;
    e[1] = (float)(s / (double)cf_);
    e[0] = ratio ? (float)((double)cf_ / c_ref) : cf_;
}
static double tan_c(double th)
{
    double c = cos(th);
    if (fabs(c) < 9.3e-10)
        c = (c < 0 ? -1.0 : 1.0) * 9.313225746154785e-10;
    return (double)(float)c;
}

int main(int argc, char **argv)
{
    const char *var = argc > 1 ? argv[1] : "check";
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double PI2 = 6.283185307179586476925286766559;
    // per group g and lane: base angle b = -2 pi (lane + 64 g) / 4096 (the kernel's pass-1 twiddles)
    std::vector<float> told(2 * 4 * 15 * 64), tnew(2 * 4 * 15 * 64);
    for (int g = 0; g < 4; g++)
        for (int lane = 0; lane < 64; lane++)
        {
            const double th = -PI2 * (double)(lane + 64 * g) / 4096.0;
            for (int j = 1; j < 16; j++)
            {
                told[2 * ((g * 15 + j - 1) * 64 + lane) + 0] = (float)cos(th * j);
                told[2 * ((g * 15 + j - 1) * 64 + lane) + 1] = (float)sin(th * j);
            }
            float *e = &tnew[2 * ((g * 15) * 64 + lane)];
            tan_entry(4 * th, e + 2 * 0 * 64, 1.0, false);
            tan_entry(8 * th, e + 2 * 1 * 64, 1.0, false);
            tan_entry(12 * th, e + 2 * 2 * 64, tan_c(4 * th), true);
            for (int q = 0; q < 4; q++)
            {
                const double w16 = -PI2 / 16.0;
                tan_entry(1 * (th + w16 * q), e + 2 * (3 + 3 * q) * 64, 1.0, false);
                tan_entry(2 * (th + w16 * q), e + 2 * (4 + 3 * q) * 64, 1.0, false);
                tan_entry(3 * (th + w16 * q), e + 2 * (5 + 3 * q) * 64, tan_c(1 * (th + w16 * q)), true);
            }
        }
    cf *d_old, *d_new, *d_in, *d_out, *d_out2;
    CHECK(hipMalloc(&d_old, told.size() * 4));
    CHECK(hipMalloc(&d_new, tnew.size() * 4));
    CHECK(hipMemcpy(d_old, told.data(), told.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_new, tnew.data(), tnew.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hin(2 * 64 * 512);
    srand(1);
    for (auto &v : hin)
        v = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMalloc(&d_in, hin.size() * 4));
    CHECK(hipMemcpy(d_in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)ncu * 64 * 512 * 8));
    CHECK(hipMalloc(&d_out2, (size_t)ncu * 64 * 512 * 8));
    if (!strcmp(var, "check"))
    {
        // group 3 of the tables (the largest angles), against a float64 evaluation
        k_check<<<1, 512>>>(d_old + 3 * 15 * 64, d_new + 3 * 15 * 64, d_in, d_out, d_out2);
        CHECK(hipDeviceSynchronize());
        std::vector<float> a(2 * 16 * 512), b(2 * 16 * 512);
        CHECK(hipMemcpy(a.data(), d_out, a.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), d_out2, b.size() * 4, hipMemcpyDeviceToHost));
        double e_old = 0, e_new = 0, nrm = 0;
        for (int t = 0; t < 512; t++)
        {
            const int lane = t & 63;
            const double th = -PI2 * (double)(lane + 64 * 3) / 4096.0;
            for (int k = 0; k < 16; k++)
            {
                double re = 0, im = 0;
                for (int j = 0; j < 16; j++)
                {
                    const double ang = th * j - PI2 * j * k / 16.0;
                    const double xr = hin[2 * (j * 512 + t)], xi = hin[2 * (j * 512 + t) + 1];
                    re += xr * cos(ang) - xi * sin(ang);
                    im += xr * sin(ang) + xi * cos(ang);
                }
                const double ao = hypot(a[2 * (k * 512 + t)] - re, a[2 * (k * 512 + t) + 1] - im);
                const double an = hypot(b[2 * (k * 512 + t)] - re, b[2 * (k * 512 + t) + 1] - im);
                e_old += ao * ao;
                e_new += an * an;
                nrm += re * re + im * im;
            }
        }
        printf("check: rel l2 error old %.3g new %.3g\n", sqrt(e_old / nrm), sqrt(e_new / nrm));
        return (sqrt(e_new / nrm) < 5e-7) ? 0 : 1;
    }
    const int iters = 4000;
    auto launch = [&]() {
        const cf *tab = !strcmp(var, "new") ? d_new : d_old;
        if (!strcmp(var, "add")) hipLaunchKernelGGL(k_energy<V_ADD>, dim3(ncu), dim3(512), LDS_BYTES, 0, tab, d_in, d_out, iters);
        else if (!strcmp(var, "mul")) hipLaunchKernelGGL(k_energy<V_MUL>, dim3(ncu), dim3(512), LDS_BYTES, 0, tab, d_in, d_out, iters);
        else if (!strcmp(var, "fma")) hipLaunchKernelGGL(k_energy<V_FMA>, dim3(ncu), dim3(512), LDS_BYTES, 0, tab, d_in, d_out, iters);
        else if (!strcmp(var, "old")) hipLaunchKernelGGL(k_energy<V_OLD>, dim3(ncu), dim3(512), LDS_BYTES, 0, tab, d_in, d_out, iters);
        else if (!strcmp(var, "new")) hipLaunchKernelGGL(k_energy<V_NEW>, dim3(ncu), dim3(512), LDS_BYTES, 0, tab, d_in, d_out, iters);
        else if (!strcmp(var, "scale")) hipLaunchKernelGGL(k_energy<V_SCALE>, dim3(ncu), dim3(512), LDS_BYTES, 0, tab, d_in, d_out, iters);
        else { printf("unknown variant %s\n", var); exit(2); }
    };
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_energy<V_ADD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_energy<V_MUL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_energy<V_FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_energy<V_OLD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_energy<V_NEW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_energy<V_SCALE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    double ms_total = 0;
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs)
    {
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 8; k++)
            launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms;
        launches += 8;
    }
    // groups per launch: ncu workgroups x 8 waves x iters x 4 groups (a group = one 16-register transform of a wave)
    const double groups = (double)ncu * 8 * iters * 4;
    printf("%s: %.4f ms per launch, %.1f G wave-groups/s (%ld launches, %d CUs)\n", var, ms_total / launches,
           groups / (ms_total / launches * 1e-3) * 1e-9, launches, ncu);
    return 0;
}
