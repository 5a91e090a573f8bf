// if_fir_fft.hip — overlap-save FFT-FIR for gfx950 (BUILD-DEFINED: the reference holds no filter code,
// /root/reference/util/if-bandpass-filter/schematic.svg:174-222 is an analog LC drawing).
//
// One WAVE = one 4096-point complex FFT held entirely in registers (64 lanes x 64 points), N = 16 x 16 x 16:
//   load      reg[row] = x[64*row + lane]           (coalesced 512-byte rows straight from HBM; no LDS staging)
//   pass 1    4 x FFT16 over n0 (register-local)    + twiddle W4096^((lane+64*rho)*k0)      (table in LDS)
//   exch 1    v_permlane32_swap + v_permlane16_swap (4-lane all-to-all in registers, no LDS)
//   pass 2    4 x FFT16 over n1                     + twiddle W256^(n2*k1)
//   exch 2    16x16 transposition inside each 16-lane row through a 8.7 KB wave-private LDS buffer, 4 rounds
//   pass 3    4 x FFT16 over n2
//   multiply by H = FFT(taps)/4096 (pre-permuted table in LDS), then the mirror-image inverse; the result lands in
//   the load layout, the first 64*OVL_ROWS outputs of each block are discarded (overlap-save).
// Index algebra: tools/fft_model.py (checked against numpy.fft).  No workgroup barriers after the table load: the 8
// waves of a 512-thread workgroup are independent; FFT blocks are handed out through an atomic queue.
// FP32 VALU only (v_add/v_fma/v_pk_*), no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include "if_fir_kernels.h"

namespace if_fir
{

struct cf
{
    float x, y;
};
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f4v_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
// a * (wr + i*wi), or a * conj(w) when CONJ
template <bool CONJ>
__device__ __forceinline__ cf cmul(cf a, float wr, float wi)
{
    if (CONJ)
        wi = -wi;
    return {a.x * wr - a.y * wi, a.x * wi + a.y * wr};
}
// multiply by -j (forward) or +j (inverse)
template <bool INV>
__device__ __forceinline__ cf rot(cf a)
{
    return INV ? cf{-a.y, a.x} : cf{a.y, -a.x};
}

template <bool INV>
__device__ __forceinline__ void bfly4(cf a, cf b, cf c, cf d, cf &u0, cf &u1, cf &u2, cf &u3)
{
    const cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = rot<INV>(csub(b, d));
    u0 = cadd(t0, t2);
    u1 = cadd(t1, t3);
    u2 = csub(t0, t2);
    u3 = csub(t1, t3);
}

// 16-point FFT, natural order in and out (radix-4 x radix-4 DIF; the digit reversal is register renaming)
template <bool INV>
__device__ __forceinline__ void fft16(cf (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    cf y[4][4]; // y[q][i]
#pragma unroll
    for (int i = 0; i < 4; i++)
        bfly4<INV>(v[i], v[i + 4], v[i + 8], v[i + 12], y[0][i], y[1][i], y[2][i], y[3][i]);
    // twiddles W16^(i*q), W16 = exp(-2*pi*j/16) (conjugated for the inverse)
    y[1][1] = cmul<INV>(y[1][1], C1, -S1);
    y[2][1] = cmul<INV>(y[2][1], R, -R);
    y[3][1] = cmul<INV>(y[3][1], S1, -C1);
    y[1][2] = cmul<INV>(y[1][2], R, -R);
    y[2][2] = rot<INV>(y[2][2]);
    y[3][2] = cmul<INV>(y[3][2], -R, -R);
    y[1][3] = cmul<INV>(y[1][3], S1, -C1);
    y[2][3] = cmul<INV>(y[2][3], -R, -R);
    y[3][3] = cmul<INV>(y[3][3], -C1, S1);
#pragma unroll
    for (int q = 0; q < 4; q++)
        bfly4<INV>(y[q][0], y[q][1], y[q][2], y[q][3], v[q], v[q + 4], v[q + 8], v[q + 12]);
}

__device__ __forceinline__ void swap32(cf &vdst, cf &src)
{
    u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(vdst.x), __float_as_uint(src.x), false, false);
    vdst.x = __uint_as_float(r.x);
    src.x = __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(vdst.y), __float_as_uint(src.y), false, false);
    vdst.y = __uint_as_float(r.x);
    src.y = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap16(cf &vdst, cf &src)
{
    u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(vdst.x), __float_as_uint(src.x), false, false);
    vdst.x = __uint_as_float(r.x);
    src.x = __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(vdst.y), __float_as_uint(src.y), false, false);
    vdst.y = __uint_as_float(r.x);
    src.y = __uint_as_float(r.y);
}

// physical register slot of logical element (i, j): j = n1 / k1 / n2 / k2 of group i (see tools/fft_model.py)
__device__ __forceinline__ constexpr int phys(int i, int j)
{
    return 4 * (i + 4 * (j & 1) + 8 * ((j >> 1) & 1)) + (j >> 2);
}

constexpr int FFT_N = 4096;
constexpr int XROW = 136;             // bytes per 16-entry row of the exchange-2 buffer (16*8 + 8 pad)
constexpr int XBUF = 4 * 16 * XROW;   // per-wave exchange buffer
constexpr int FFT_WAVES = 8;
constexpr int LDS_TW1 = 0, LDS_HP = 32768, LDS_TW2 = 65536, LDS_XB = 65536 + 2048;
constexpr int FFT_LDS_BYTES = LDS_XB + FFT_WAVES * XBUF;

__device__ __forceinline__ void exchange1_fwd(cf (&r)[64])
{
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++)
#pragma unroll
        for (int rho = 0; rho < 4; rho++)
            swap32(r[4 * k0 + rho], r[4 * (k0 + 8) + rho]);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rho = 0; rho < 4; rho++)
                swap16(r[4 * (i + 8 * b) + rho], r[4 * (i + 4 + 8 * b) + rho]);
}
__device__ __forceinline__ void exchange1_inv(cf (&r)[64])
{
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rho = 0; rho < 4; rho++)
                swap16(r[4 * (i + 8 * b) + rho], r[4 * (i + 4 + 8 * b) + rho]);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++)
#pragma unroll
        for (int rho = 0; rho < 4; rho++)
            swap32(r[4 * k0 + rho], r[4 * (k0 + 8) + rho]);
}

// 16x16 transposition inside each 16-lane row: element (i, j) of lane (g, m) -> lane (g, j), slot (i, m)
__device__ __forceinline__ void exchange2(cf (&r)[64], char *xb, int lane)
{
    const int g = lane >> 4, m = lane & 15;
    char *wr = xb + g * (16 * XROW) + m * 8;      // + j*XROW : element (j, m)
    const char *rd = xb + g * (16 * XROW) + m * XROW; // + j*8  : element (m, j)
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
#pragma unroll
        for (int j = 0; j < 16; j++)
            *reinterpret_cast<f2v *>(wr + j * XROW) = (f2v){r[phys(i, j)].x, r[phys(i, j)].y};
#pragma unroll
        for (int j = 0; j < 16; j++)
        {
            const f2v t = *reinterpret_cast<const f2v *>(rd + j * 8);
            r[phys(i, j)] = {t.x, t.y};
        }
    }
}

template <int OVL_ROWS>
__global__ __launch_bounds__(512, 2) void fir_fft_kernel(const f2v *__restrict__ in, f2v *__restrict__ out,
                                                        const f2v *__restrict__ tables, const f2v *__restrict__ hist,
                                                        int T, int64_t N, int64_t nblocks, int32_t waves_total,
                                                        unsigned int *queue)
{
    constexpr int OVL = 64 * OVL_ROWS;
    constexpr int L = FFT_N - OVL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- tables: global -> LDS (once per workgroup) ------------------------------------------------------------
    {
        const f4v_t *src = reinterpret_cast<const f4v_t *>(tables);
        f4v_t *dst = reinterpret_cast<f4v_t *>(smem);
        for (int i = threadIdx.x; i < LDS_XB / 16; i += 512)
            dst[i] = src[i];
    }
    __syncthreads();
    const f2v *tw1 = reinterpret_cast<const f2v *>(smem + LDS_TW1);
    const f2v *hp = reinterpret_cast<const f2v *>(smem + LDS_HP);
    const f2v *tw2 = reinterpret_cast<const f2v *>(smem + LDS_TW2);
    char *xb = smem + LDS_XB + wid * XBUF;

    int64_t blk = (int64_t)blockIdx.x * FFT_WAVES + wid;
    unsigned int ticket = 0;
    while (blk < nblocks)
    {
        if (lane == 0)
            ticket = atomicAdd(queue, 1u);   // next block for this wave, resolved at the end of the iteration
        const int64_t s0 = blk * L - OVL;     // stream index of the block's first sample
        cf r[64];
        if (s0 >= 0 && s0 + FFT_N <= N)
        {
            const f2v *src = in + s0 + lane;
#pragma unroll
            for (int row = 0; row < 64; row++)
            {
                const f2v t = src[row * 64];
                r[row] = {t.x, t.y};
            }
        }
        else
        {
#pragma unroll
            for (int row = 0; row < 64; row++)
            {
                const int64_t gidx = s0 + row * 64 + lane;
                f2v t = {0.f, 0.f};
                if (gidx >= 0)
                {
                    if (gidx < N)
                        t = in[gidx];
                }
                else if (gidx >= -(int64_t)(T - 1))
                    t = hist[(T - 1) + gidx];
                r[row] = {t.x, t.y};
            }
        }

        // ---- forward ------------------------------------------------------------------------------------------
#pragma unroll
        for (int rho = 0; rho < 4; rho++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
                t[j] = r[4 * j + rho];
            fft16<false>(t);
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                if (j == 0)
                    r[4 * j + rho] = t[j];
                else
                {
                    const f2v w = tw1[(rho * 16 + j) * 64 + lane];
                    r[4 * j + rho] = cmul<false>(t[j], w.x, w.y);
                }
            }
        }
        exchange1_fwd(r);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
                t[j] = r[phys(i, j)];
            fft16<false>(t);
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                if (j == 0)
                    r[phys(i, j)] = t[j];
                else
                {
                    const f2v w = tw2[j * 16 + (lane & 15)];
                    r[phys(i, j)] = cmul<false>(t[j], w.x, w.y);
                }
            }
        }
        exchange2(r, xb, lane);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
                t[j] = r[phys(i, j)];
            fft16<false>(t);
            // ---- pointwise multiply by H/4096 and start the inverse (pass 3^-1) in the same registers -----------
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                const f2v h = hp[(i * 16 + j) * 64 + lane];
                t[j] = cmul<false>(t[j], h.x, h.y);
            }
            fft16<true>(t);
#pragma unroll
            for (int j = 0; j < 16; j++)
                r[phys(i, j)] = t[j];
        }
        exchange2(r, xb, lane);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                if (j == 0)
                    t[j] = r[phys(i, j)];
                else
                {
                    const f2v w = tw2[j * 16 + (lane & 15)];
                    t[j] = cmul<true>(r[phys(i, j)], w.x, w.y);
                }
            }
            fft16<true>(t);
#pragma unroll
            for (int j = 0; j < 16; j++)
                r[phys(i, j)] = t[j];
        }
        exchange1_inv(r);
#pragma unroll
        for (int rho = 0; rho < 4; rho++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                if (j == 0)
                    t[j] = r[4 * j + rho];
                else
                {
                    const f2v w = tw1[(rho * 16 + j) * 64 + lane];
                    t[j] = cmul<true>(r[4 * j + rho], w.x, w.y);
                }
            }
            fft16<true>(t);
#pragma unroll
            for (int j = 0; j < 16; j++)
                r[4 * j + rho] = t[j];
        }

        // ---- store the valid part --------------------------------------------------------------------------------
        const int64_t o0 = blk * L + lane;
        if (blk * L + L <= N)
        {
#pragma unroll
            for (int row = OVL_ROWS; row < 64; row++)
                out[o0 + (row - OVL_ROWS) * 64] = (f2v){r[row].x, r[row].y};
        }
        else
        {
#pragma unroll
            for (int row = OVL_ROWS; row < 64; row++)
                if (o0 + (row - OVL_ROWS) * 64 < N)
                    out[o0 + (row - OVL_ROWS) * 64] = (f2v){r[row].x, r[row].y};
        }
        blk = (int64_t)waves_total + (int64_t)__builtin_amdgcn_readfirstlane(ticket);
    }
}

template <int OVL_ROWS>
static hipError_t launch_fft_t(const LaunchArgs &a)
{
    auto kern = fir_fft_kernel<OVL_ROWS>;
    constexpr int L = FFT_N - 64 * OVL_ROWS;
    static bool attr_done[16] = {false};
    static int cus[16] = {0};
    const int dev = a.device & 15;
    if (!attr_done[dev])
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, FFT_LDS_BYTES);
        if (e != hipSuccess)
            return e;
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, a.device);
        if (e != hipSuccess)
            return e;
        cus[dev] = prop.multiProcessorCount;
        attr_done[dev] = true;
    }
    const int64_t nblocks = (a.N + L - 1) / L;
    if (nblocks <= 0)
        return hipSuccess;
    int64_t wgs = cus[dev];
    if (wgs * FFT_WAVES > nblocks)
        wgs = (nblocks + FFT_WAVES - 1) / FFT_WAVES;
    hipError_t e = hipMemsetAsync(a.queue, 0, 16, a.stream);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(512), FFT_LDS_BYTES, a.stream,
                       reinterpret_cast<const f2v *>(a.in), reinterpret_cast<f2v *>(a.out),
                       reinterpret_cast<const f2v *>(a.fft_tables), reinterpret_cast<const f2v *>(a.hist), a.T, a.N,
                       nblocks, (int32_t)(wgs * FFT_WAVES), (unsigned int *)a.queue);
    return hipGetLastError();
}

bool fft_supported(int T, int D)
{
    return D == 1 && T >= 1 && T <= 1025;
}

int fft_overlap_rows(int T)
{
    return (T - 1 <= 256) ? 4 : 16;
}

hipError_t launch_fft(const LaunchArgs &a)
{
    if (!fft_supported(a.T, a.D) || !a.fft_tables)
        return hipErrorInvalidConfiguration;
    return fft_overlap_rows(a.T) == 4 ? launch_fft_t<4>(a) : launch_fft_t<16>(a);
}

// Host side: twiddle and H tables in the kernel's LDS image order (float64 math, rounded once to float32).
//   [0, 32 KB)      tw1[(rho*16+k0)*64 + lane] = W4096^((lane+64*rho)*k0)
//   [32 KB, 64 KB)  hp [(i*16+k2)*64 + lane]   = FFT(taps)[(4*(lane/16)+i) + 16*(lane%16) + 256*k2] / 4096
//   [64 KB, 66 KB)  tw2[k1*16 + n2]            = W256^(n2*k1)
void fft_build_tables(const float *taps, int T, float *tables /* FFT_TABLE_FLOATS floats */)
{
    const double PI2 = 6.283185307179586476925286766559;
    float *tw1 = tables, *hp = tables + 2 * 4096, *tw2 = tables + 4 * 4096;
    for (int rho = 0; rho < 4; rho++)
        for (int k0 = 0; k0 < 16; k0++)
            for (int lane = 0; lane < 64; lane++)
            {
                const int e = ((lane + 64 * rho) * k0) % 4096;
                const double a = -PI2 * (double)e / 4096.0;
                tw1[2 * ((rho * 16 + k0) * 64 + lane) + 0] = (float)cos(a);
                tw1[2 * ((rho * 16 + k0) * 64 + lane) + 1] = (float)sin(a);
            }
    for (int k1 = 0; k1 < 16; k1++)
        for (int n2 = 0; n2 < 16; n2++)
        {
            const double a = -PI2 * (double)((n2 * k1) % 256) / 256.0;
            tw2[2 * (k1 * 16 + n2) + 0] = (float)cos(a);
            tw2[2 * (k1 * 16 + n2) + 1] = (float)sin(a);
        }
    // DFT of the taps in float64 with an exact-argument table
    static double ct[4096], st[4096];
    for (int e = 0; e < 4096; e++)
    {
        ct[e] = cos(-PI2 * (double)e / 4096.0);
        st[e] = sin(-PI2 * (double)e / 4096.0);
    }
    for (int i = 0; i < 4; i++)
        for (int k2 = 0; k2 < 16; k2++)
            for (int lane = 0; lane < 64; lane++)
            {
                const int k = (4 * (lane / 16) + i) + 16 * (lane % 16) + 256 * k2;
                double re = 0.0, im = 0.0;
                for (int n = 0; n < T; n++)
                {
                    const int e = (int)(((int64_t)k * n) & 4095);
                    re += (double)taps[n] * ct[e];
                    im += (double)taps[n] * st[e];
                }
                hp[2 * ((i * 16 + k2) * 64 + lane) + 0] = (float)(re / 4096.0);
                hp[2 * ((i * 16 + k2) * 64 + lane) + 1] = (float)(im / 4096.0);
            }
}

} // namespace if_fir
