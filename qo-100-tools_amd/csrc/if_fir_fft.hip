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
//
// Build: this file is compiled seven times (csrc/Makefile) -- once per overlap length with -DIF_FIR_FFT_ROWS=4|8|16|32|48 (the
// kernel, its launcher and the explicit instantiation of launch_fft_rows<ROWS>; the 32-row unit also carries the two-partition
// launches), once with -DIF_FIR_FFT_ODD (the odd-decimation kernel) and once with neither (host side: tables, routing predicates,
// launch_fft) -- so that the instantiations compile
// in parallel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <type_traits>

#include <vector>

#include "if_fir_kernels.h"
#include "if_fir_fft_queue.h"

// the host side (tables, routing predicates, launch_fft) is the unit compiled without a kernel selector
#if !defined(IF_FIR_FFT_ROWS) && !defined(IF_FIR_FFT_ODD)
#define IF_FIR_FFT_HOST 1
#endif

namespace if_fir
{

// complex = one aligned VGPR pair (re, im): adds are single v_pk_add_f32, a complex multiply is v_pk_mul_f32 +
// v_pk_fma_f32 with the swap/negate folded into op_sel / neg modifiers
typedef float cf __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f4v_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }

// Complex multiply a * w (or a * conj(w)) in two packed instructions, swap/negate folded into VOP3P modifiers:
//   t = a * (w.x, w.x);   d = (a.y, a.x) * (-+w.y, +-w.y) + t
template <bool CONJ>
__device__ __forceinline__ cf cmul_v(cf a, cf w) // w in a VGPR pair (LDS tables)
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
__device__ __forceinline__ cf cmul_s(cf a, cf w) // w wave-uniform (compile-time twiddle) in an SGPR pair
{
    cf t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    if (CONJ)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}
// acc + a * w in two packed FMAs (w in a VGPR pair)
__device__ __forceinline__ cf cmac_v(cf acc, cf a, cf w)
{
    cf t, d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
// a + w*b and a - w*b with w = -j (forward) or +j (inverse): one v_pk_add_f32 each
template <bool INV>
__device__ __forceinline__ cf add_rot(cf a, cf b)
{
    cf d;
    if (INV) // (a.x - b.y, a.y + b.x)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else     // (a.x + b.y, a.y - b.x)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <bool INV>
__device__ __forceinline__ cf sub_rot(cf a, cf b)
{
    return add_rot<!INV>(a, b);
}

template <bool INV>
__device__ __forceinline__ void bfly4(cf a, cf b, cf c, cf d, cf &u0, cf &u1, cf &u2, cf &u3)
{
    const cf t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    u0 = t0 + t2;
    u1 = add_rot<INV>(t1, t3);
    u2 = t0 - t2;
    u3 = sub_rot<INV>(t1, t3);
}
// same with input c pre-multiplied by -j/+j (folded into the first adds)
template <bool INV>
__device__ __forceinline__ void bfly4_crot(cf a, cf b, cf c, cf d, cf &u0, cf &u1, cf &u2, cf &u3)
{
    const cf t0 = add_rot<INV>(a, c), t1 = sub_rot<INV>(a, c), t2 = b + d, t3 = b - d;
    u0 = t0 + t2;
    u1 = add_rot<INV>(t1, t3);
    u2 = t0 - t2;
    u3 = sub_rot<INV>(t1, t3);
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
    // twiddles W16^(i*q), W16 = exp(-2*pi*j/16) (conjugated for the inverse); W16^4 = -j is folded into stage 2
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

// 8-point FFT, natural order in and out (one radix-2 stage with twiddles W8^a, two radix-4 butterflies): 28 packed instructions
template <bool INV>
__device__ __forceinline__ void fft8(cf (&v)[8])
{
    constexpr float R = 0.70710678118654752f;
    cf u[4], d[4];
#pragma unroll
    for (int a = 0; a < 4; a++)
    {
        u[a] = v[a] + v[a + 4];
        d[a] = v[a] - v[a + 4];
    }
    d[1] = cmul_s<INV>(d[1], (cf){R, -R});
    d[3] = cmul_s<INV>(d[3], (cf){-R, -R});
    bfly4<INV>(u[0], u[1], u[2], u[3], v[0], v[2], v[4], v[6]);
    bfly4_crot<INV>(d[0], d[1], d[2], d[3], v[1], v[3], v[5], v[7]); // (d[2] carries W8^2 = -j: folded into the butterfly's adds)
}

// ---- twiddles in (cos, tan) form (round 4) ---------------------------------------------------------------------------------
// A twiddle w = c (1 + j t) is kept as the pair E = (c, t).  x (1 + j t) is ONE packed FMA and a +- c u another, so a radix-4
// butterfly whose inputs 1..3 carry twiddles w1, w2, w3 is 11 packed FMAs (3 twiddle multiplies + 8 adds = 14 instructions in the
// usual form); the third entry of a butterfly holds (c3 / c1, t3):
//   u2 = x2 (1 + j t2);  t0 = x0 + c2 u2;  t1 = x0 - c2 u2;  u1, u3 likewise;  v+- = u1 +- (c3 / c1) u3;
//   X0 = t0 + c1 v+;  X2 = t0 - c1 v+;  X1 = t1 -+ j c1 v-;  X3 = t1 +- j c1 v-
// A 16-point transform whose input j carries b^j (b = the lane's base twiddle: the twiddle a three-pass transform applies
// between two passes, moved from the outputs of one pass to the inputs of the next) is 8 such butterflies = 88 packed
// instructions where transform + 15 twiddle multiplies were 110.  An exact zero of a cosine is stored as 2^-30 (host,
// tan_entry): the tangent stays finite and the products are exact to rounding.  Measured (tools/ubench_energy.hip,
// profiles/r04_energy_per_instruction.txt): a packed FMA costs 1.18 x the energy of a packed add; the group as a whole -4.5 %.
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
    X1 = tw_ajc<INV>(t1, vm, e1); // forward: t1 - j c1 v-
    X3 = tw_ajc<!INV>(t1, vm, e1);
}
// 16-point transform of v[j] b^j (inverse: v[j] conj(b)^j), natural order in and out.  Table (host, tan_fft16_entries): entries
// 0..2 = b^4, b^8, b^12 (first radix-4 stage; its outputs still owe b^i); entries 3 + 3 q + (i - 1) = b^i W16^(i q), i = 1..3 (the
// owed factor merged with the transform's own twiddle); entry k at e[k * STRIDE].
template <bool INV, int STRIDE>
__device__ __forceinline__ void fft16_tw(cf (&v)[16], const f2v *e)
{
    cf y[4][4];
    {
        const cf e1 = e[0], e2 = e[STRIDE], e3 = e[2 * STRIDE];
#pragma unroll
        for (int i = 0; i < 4; i++)
            bfly4_tw<INV>(v[i], v[i + 4], v[i + 8], v[i + 12], e1, e2, e3, y[0][i], y[1][i], y[2][i], y[3][i]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        bfly4_tw<INV>(y[q][0], y[q][1], y[q][2], y[q][3], e[(3 + 3 * q) * STRIDE], e[(4 + 3 * q) * STRIDE], e[(5 + 3 * q) * STRIDE],
                      v[q], v[q + 4], v[q + 8], v[q + 12]);
}

// The same with the table in two pieces (round 4, full-rate pipeline): the first stage's entries at s1[0], s1[S1], s1[2 S1]; the
// second stage's from the SHARED table T of the triples (b, b^2, b^3 with the third as (c3 / c1, t3)) of b = W4096^m, m = 0..1023:
// entry (b W16^q)^(j+1) = T_j[m + 256 q] at tq[1024 j + 256 q] (three arrays of 1024 entries; tq = T + tsw(m)).
template <bool INV, int S1>
__device__ __forceinline__ void fft16_tw_T(cf (&v)[16], const f2v *s1, const f2v *tq)
{
    cf y[4][4];
    {
        const cf e1 = s1[0], e2 = s1[S1], e3 = s1[2 * S1];
#pragma unroll
        for (int i = 0; i < 4; i++)
            bfly4_tw<INV>(v[i], v[i + 4], v[i + 8], v[i + 12], e1, e2, e3, y[0][i], y[1][i], y[2][i], y[3][i]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        bfly4_tw<INV>(y[q][0], y[q][1], y[q][2], y[q][3], tq[256 * q], tq[1024 + 256 * q], tq[2048 + 256 * q], v[q], v[q + 4], v[q + 8],
                      v[q + 12]);
}
// Position of entry m in an array of T: the low five bits are mixed with bits 5..7 so that both users' gathers -- m = lane + 64 rho
// + 256 q (last inverse pass) and m = 4 (lane / 16) + i + 16 (lane % 16) + 256 q (forward pass 3) -- put the 32 lanes of a half
// wave on 32 different 8-byte bank slots (checked exhaustively by tests/test_host.py); bits 8, 9 are untouched: + 256 q stays an
// offset.  Host twin: fft_tsw.
__host__ __device__ __forceinline__ constexpr unsigned tsw(unsigned m)
{
    return (m & ~31u) | ((m ^ (m >> 5)) & 1u) | (((m >> 1) ^ (m >> 6)) & 1u) << 1 | (m & 4u) | (((m >> 3) ^ (m >> 7)) & 1u) << 3 | (m & 16u);
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
constexpr int FFT_PART = 2048; // filters of 3074..4096 taps: two partitions of at most this many taps
constexpr int XROW = 136;             // bytes per 16-entry row of the exchange buffers (16*8 + 8 pad)
constexpr int XREG = 16 * XROW + 32;  // one 16x16 region (+32 so that the 4 regions start on different banks)
constexpr int XBUF = 4 * XREG;        // per-wave exchange buffer
constexpr int FFT_WAVES = 8;
static_assert(FFT_WAVES == (int)QB, "one slot of a block group per wave of the workgroup");
constexpr int LDS_TW1 = 0, LDS_HP = 32768, LDS_TW2 = 65536, LDS_TWD = 65536 + 2048, LDS_TWE = LDS_TWD + 8192,
              LDS_NCO = LDS_TWE + 8192, LDS_TWF = LDS_NCO + 512, LDS_XB = LDS_TWF + 2048;
static_assert(LDS_XB == FFT_TABLE_FLOATS * 4, "table image size");
// Image of the decimate-by-4 kernels (round 4, twiddles in (cos, tan) form; same size, other contents -- fft_build_tables):
//   LDS_TW1: pass 3, first stage   [(i*3 + e)*64 + lane]   b = W4096^(k0 + 16 k1), k0 = 4 (lane/16) + i, k1 = lane%16
//   LDS_TW2: pass 2                [(i*15 + e)*4 + lane/16] b = W256^k0
//   LDS_TWD: inverse, last pass    [e*64 + lane]            b = W1024^lane
//   LDS_TWE: inverse, middle pass  [e*4 + lane%4]           b = W64^(lane%4)
//   LDS_HP : G'[m0][q] = b^m0 G[m0][q] (the factor pass 3's first stage still owes, merged into the table)
// Image of the full-rate pipeline (D = 1, the selecting store, their accumulating forms; same size again):
//   LDS_TW1: [0, 6 KB) forward pass 3, first stage, as above; [8 KB, 32 KB) T: three arrays of 1024 entries (fft16_tw_T, tsw)
//   LDS_TW2: forward pass 2 as above;  LDS_TWD: inverse pass 2 [e*16 + lane%16], b = W256^(lane%16);  LDS_HP: H / 4096
constexpr int LDS_TT = LDS_TW1 + 8192;
// Phasor tables (round 5): every (cos, tan) image leaves bytes [6 KB, 8 KB) of the LDS_TW1 slot free; they hold P1[k] = exp(j 2 pi k /
// 2^7) and P2[k] = exp(j 2 pi k / 2^14), k = 0..127 (host, fft_phasor_tables), and a 32-bit phase becomes a phasor with two table
// reads, a second-order polynomial for its low 18 bits (angle < 3.9e-4 rad: the cubic term is 1e-11) and two complex multiplies --
// about 12 instructions where the two sincospif of nco_phasor are about 80, once per block and lane in every kernel with an NCO and
// once per channel group in the filter bank's general forms.
constexpr int LDS_PH = LDS_TW1 + 6144;
__device__ __forceinline__ cf lds_phasor(const f2v *pht, uint32_t ph) // exp(+j 2 pi ph / 2^32)
{
    const cf a = pht[ph >> 25], b = pht[128u + ((ph >> 18) & 127u)];
    const float th = (float)(ph & 0x3ffffu) * 1.4629180792671596e-9f; // 2 pi / 2^32
    const cf lo = {__builtin_fmaf(-0.5f * th, th, 1.0f), th};
    return cmul_v<false>(cmul_v<false>(a, b), lo);
}
// Row loads: the first and last EDGE rows of a block keep the default cache policy, the rows in between are `nt`.  EDGE = the block
// overlap (the neighbouring block finds the shared rows in L2, round 2).  Round 4 swept larger values (IF_FIR_FFT_EDGE_MIN_FULL /
// _DEC for the full-rate pipeline / the decimating tails, profiles/r04_edge_rows.txt): 2^28-sample launches lose 2-3 % with more
// cached rows; configs[1] (2^26 samples) GAINS 4.5 % at 16 rows each side -- half of its 512 MB input, i.e. the 256 MB
// memory-side cache serving the same bytes again on the benchmark's next launch: an artefact of re-filtering one buffer, not a
// property of a stream in service, so it was not adopted.
#ifndef IF_FIR_FFT_EDGE_MIN_FULL
#define IF_FIR_FFT_EDGE_MIN_FULL 0
#endif
#ifndef IF_FIR_FFT_EDGE_MIN_DEC
#define IF_FIR_FFT_EDGE_MIN_DEC 0
#endif
#ifndef IF_FIR_FFT_TAN
#define IF_FIR_FFT_TAN 1 // 0: the decimate-by-4 kernels in round 3's form (A/B builds)
#endif
// Kernel argument of the tails: the filter-bank forms (CHAN >= 4) take the whole ChanArgs (2.4 KB by value), the single-channel
// kernels only the thinning factor -- the headline path's launches then copy 150 bytes of kernel arguments instead of 2.5 KB
struct ChanNone
{
    uint32_t sub;
};
template <int CHAN>
using chan_arg_t = typename std::conditional<(CHAN >= 4), ChanArgs, ChanNone>::type;
constexpr int LDS_Q = LDS_XB + FFT_WAVES * XBUF; // workgroup block queue: slot counter (16 B) + ring of group entries
constexpr int LDS_QPTR = LDS_Q + 16 + Q_RING * 8; // 16-slot bank: the 16 output pointers (kept out of the SGPRs)
constexpr int LDS_QNCO = LDS_QPTR + 16 * 8; // bank tails with an NCO: the block's rotation phasor, one 8-byte word per wave
// tail phase of the queue (short launches): tail word (8 B) and the four SIMDs' claim counters (4 x 4 B)
constexpr int LDS_QTAIL = LDS_QNCO + FFT_WAVES * 8, LDS_QCLAIM = LDS_QTAIL + 16;
// filter bank at decimation 8 (round 4): W16^(a s), s = 0..15, a = 0..7 (1 KB) and per channel the 16 row phasors of its mix-down
// (CHAN_MAX x 16 entries), both computed by the workgroup at the start of the launch
constexpr int LDS_ROWT = LDS_QCLAIM + 16;
constexpr int FFT_LDS_BYTES = LDS_ROWT + CHAN_MAX * 16 * 8;
static_assert(FFT_LDS_BYTES <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");

__device__ __forceinline__ void exchange1_fwd(cf (&r)[64])
{
    asm volatile("s_nop 1"); // inline-asm VALU write -> v_permlane read needs 2 wait states
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
    asm volatile("s_nop 1");
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

// ---- lane exchanges through LDS: the order of their phases is a property of the BUILD (round 5) -----------------
// Lanes exchange data through the wave's private LDS buffer in several places below: one phase of 16 writes per lane, one of 16
// reads, then the next exchange's writes into the same buffer.  The hardware executes a wave's LDS instructions in order, but for
// the COMPILER these are plain loads and stores of ONE thread, and it may reorder a load and a store whenever it can prove that
// they never overlap.  For most pairs of these exchanges such a proof exists: in exchange2 write j goes to base + 8 m + 136 j and
// read j' comes from base + 136 m + 8 j'; the difference is 128 m + 8 (j' - j) - 128 j, i.e. 8 (j' - j) modulo 128 -- never within
// 8 bytes of 0 for j != j', exactly the variable-scale / constant-offset test of LLVM's BasicAA (only the pair j = j' = m really
// overlaps).  So read j' could legally be placed ahead of writes j' + 1 .. 15, and the next exchange's write j ahead of this one's
// reads -- and then a lane reads a slot its partner lane has not written yet, or has overwritten already.  Round 4 saw exactly that
// in the odd-decimation kernel's transposition (garbage outputs) and answered with a compiler fence there; the other exchanges
// were in order "today" and had the fence switched off because it cost 0.65 % on the headline (it pins the table reads too).
// Round 5, two measures that cost nothing at run time:
//  (1) every exchange READ goes through a base address that has passed through an empty `asm volatile` (lds_opaque): the compiler
//      knows nothing about its value, no alias-freedom proof against any LDS store exists any more, and the single-thread
//      semantics of the language pin every exchange read behind the writes before it and every later exchange write behind the
//      read -- while the table reads (plain, read-only data) stay free to move, which is what the blunt fence took away;
//  (2) the build checks the result: all exchange accesses are made by the two helpers below (xst16 / xld16); the units are compiled with
//      line tables (-gline-tables-only: no effect on the generated code), and tools/check_lds_exchange.py walks every kernel's
//      disassembly, classifies each DS instruction by its source line and fails the build unless the exchange stream is strictly
//      16 stores, 16 loads, 16 stores, ... (csrc/Makefile; tests/test_host.py compiles a deliberately mis-ordered probe,
//      -DIF_FIR_FFT_LDSX_PROBE=1, and sees it flagged).
__device__ __forceinline__ const char *lds_opaque(const char *p)
{
    // (the asm operand is the 32-bit LDS pointer itself, not an integer: an inttoptr would be re-materialised next to every load
    // by the address-sinking pass and the load-store vectorizer would no longer see one base -- no ds_read2_b64)
    const __attribute__((address_space(3))) char *q = (const __attribute__((address_space(3))) char *)p;
    asm volatile("" : "+v"(q));
    return (const char *)q;
}
// One phase of an exchange: element j at p + j STRIDE.  (The empty asm on the loaded values emits nothing; it keeps the DS
// instructions attributed to THESE lines: a value that goes straight into one of the inline-asm butterflies is otherwise
// re-created by the DAG combiner -- bitcast of a load -> load of the other type -- with the source line of that butterfly, and the
// gate could not tell the exchange load from a table read.  It stands behind all 16 loads so that the load-store vectorizer still
// pairs them into ds_read2_b64.)
template <int STRIDE>
__device__ __forceinline__ void xst16(char *p, const cf (&v)[16])
{
#pragma unroll
    for (int j = 0; j < 16; j++)
        *reinterpret_cast<f2v *>(p + j * STRIDE) = v[j]; /* LDSX:STORE (the gate keys on this line) */
}
template <int STRIDE>
__device__ __forceinline__ void xld16(const char *p, cf (&v)[16])
{
#pragma unroll
    for (int j = 0; j < 16; j++)
        v[j] = *reinterpret_cast<const f2v *>(p + j * STRIDE); /* LDSX:LOAD (the gate keys on this line) */
#pragma unroll
    for (int j = 0; j < 16; j++)
        asm("" : "+v"(v[j])); /* LDSX:LOAD (a load folded into its user takes this line) */
}
// The lane's four exchange addresses in its wave's buffer (computed once per kernel; the read bases opaque):
//   X (16x16 transposition inside each 16-lane row g; m = lane % 16): element j is written to wx + j XROW, read from rx + 8 j
//   Y (inverse_tail256 / inverse_dec4_tan: element mu1 of lane (k0, low) -> lane 4 mu1 + low, slot k0): wy + j XROW, ry + 8 j
struct XAddr
{
    char *wx;
    const char *rx;
    char *wy;
    const char *ry;
};
__device__ __forceinline__ XAddr xaddr_x(char *xb, int lane)
{
    const int g = lane >> 4, m = lane & 15;
    return XAddr{xb + g * XREG + m * 8, lds_opaque(xb + g * XREG + m * XROW), nullptr, nullptr};
}
__device__ __forceinline__ XAddr xaddr_xy(char *xb, int lane)
{
    const int g = lane >> 4, m = lane & 15;
    const int k0 = 4 * g + (m >> 2), low = m & 3;
    return XAddr{xb + g * XREG + m * 8, lds_opaque(xb + g * XREG + m * XROW), xb + low * XREG + k0 * 8,
                 lds_opaque(xb + (lane & 3) * XREG + (lane >> 2) * XROW)};
}

// 16x16 transposition inside each 16-lane row: element (i, j) of lane (g, m) -> lane (g, j), slot (i, m)
__device__ __forceinline__ void exchange2(cf (&r)[64], const XAddr &xa)
{
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        cf t[16];
#pragma unroll
        for (int j = 0; j < 16; j++)
            t[j] = r[phys(i, j)];
#if defined(IF_FIR_FFT_LDSX_PROBE) && IF_FIR_FFT_LDSX_PROBE == 1
        // (tests/test_host.py: a deliberately mis-ordered exchange -- the second half of the stores behind the loads of the first
        // half's partners; the gate must flag it)
#pragma unroll
        for (int j = 0; j < 8; j++)
            *reinterpret_cast<f2v *>(xa.wx + j * XROW) = t[j]; /* LDSX:STORE (probe) */
        cf u[16];
        xld16<8>(xa.rx, u);
#pragma unroll
        for (int j = 8; j < 16; j++)
            *reinterpret_cast<f2v *>(xa.wx + j * XROW) = t[j]; /* LDSX:STORE (probe) */
#pragma unroll
        for (int j = 0; j < 16; j++)
            t[j] = u[j];
#else
        xst16<XROW>(xa.wx, t);
        xld16<8>(xa.rx, t);
#endif
#pragma unroll
        for (int j = 0; j < 16; j++)
            r[phys(i, j)] = t[j];
    }
}

// ---- memory helpers --------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t srd_t;
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2_t;

// buffer descriptor over [p, p + bytes): wave-uniform by construction (readfirstlane) so that hipcc emits plain
// buffer_load/store with the descriptor in SGPRs (no waterfall loop); out-of-range lanes read 0 / are not written
__device__ __forceinline__ srd_t make_srd(const void *p, int64_t bytes)
{
    const uint64_t a = (uint64_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int64_t clipped = bytes < 0 ? 0 : (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes);
    const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)clipped);
    void *q = (void *)(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, n, 0x00020000);
}
// Cache policy of the streams (aux bits of the buffer instructions: 2 = nt, non-temporal).  Measured on 2^28 samples
// (profiles/r02_nt_ab.txt): nt stores help every configuration (255 taps /4: 0.502 -> 0.482 ms, the outputs are never
// read again).  Row loads: the rows a block shares with its neighbours (the first and last OVL_ROWS rows) keep the
// default policy -- the neighbouring block is being loaded by the next wave of the same workgroup at about the same
// time and finds them in L2: HBM reads 2.269 -> 2.161 GB per launch = 1.006 x algorithmic, -2 % time -- and the rows
// in between, which nobody reads again, are nt.  (nt on ALL rows costs 3 % at 16 overlap rows.)
#ifndef IF_FIR_FFT_LOAD_AUX
#define IF_FIR_FFT_LOAD_AUX(ovl_rows) 2
#endif
#ifndef IF_FIR_FFT_STORE_AUX
#define IF_FIR_FFT_STORE_AUX 2
#endif
// decimate-by-4 tail: how many of the 4 batches of next-block row loads are issued during pass 3 (the rest behind the
// small inverse).  4 fits without scratch since round 2 and measures the same (0.4546 vs 0.4549 ms): 3 is kept.
#ifndef IF_FIR_FFT_EARLY_GROUPS
#define IF_FIR_FFT_EARLY_GROUPS 3
#endif
// the first block's rows are requested ahead of the table copy (head of the launch)
#ifndef IF_FIR_FFT_TABLE_COPY_UNROLLED
#define IF_FIR_FFT_TABLE_COPY_UNROLLED 1 // table copy global -> LDS with all loads of a thread in flight (0: one at a time)
#endif
#ifndef IF_FIR_FFT_LOADS_FIRST
#define IF_FIR_FFT_LOADS_FIRST 1
#endif
template <int AUX = 0>
__device__ __forceinline__ cf buf_load(srd_t rsrc, unsigned voff, unsigned soff)
{
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, AUX);
    return (cf){__uint_as_float(v[0]), __uint_as_float(v[1])};
}
// int16 IQ front-end: one dword = (I, Q) as two int16; value = int16 * 2^-15
__device__ __forceinline__ cf cvt_i16(unsigned w)
{
    // the 2^-15 of the sample format is folded into the H table (fft_build_tables, in_scale): a power of two commutes
    // exactly with every float operation on the way, and the 64 multiplies per block are saved
    return (cf){(float)(short)(w & 0xffffu), (float)((int)w >> 16)};
}
// Row `row` of a block (sample row*64 + lane) whose descriptor starts at the block's first sample.  float32 rows land
// in r[row]; int16 rows stay RAW (one dword, kept in the register of r[row].x: the row's register pair is dead until
// pass 1 writes it, so the raw block costs no registers of its own) and are converted when pass 1 consumes them — converting at
// the load would put a vmcnt wait right behind every prefetch.
template <bool I16, int AUX>
__device__ __forceinline__ void load_row_aux(cf (&r)[64], srd_t rsrc, int lane, int row)
{
#ifdef IF_FIR_DIAG_CONTIG // (timing study builds only, results wrong: the 16 rows of a load batch are contiguous in memory)
    const int mrow = 16 * ((row >> 2) & 3) + 4 * (row >> 4) + (row & 3);
#else
    const int mrow = row;
#endif
    if constexpr (I16)
        r[row].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)lane * 4u, mrow * 256, AUX));
    else
        r[row] = buf_load<AUX>(rsrc, (unsigned)lane * 8u, mrow * 512);
}
// EDGE rows: the first and last `EDGE` rows of a block are the rows the neighbouring block shares with it; loaded with
// the default policy they are served to the neighbour from L2 (IF_FIR_FFT_EDGE_CACHED=0 switches that off for A/B runs)
#ifndef IF_FIR_FFT_EDGE_CACHED
#define IF_FIR_FFT_EDGE_CACHED 1
#endif
template <bool I16, int AUX, int EDGE = 0>
__device__ __forceinline__ void load_row(cf (&r)[64], srd_t rsrc, int lane, int row)
{
    if (IF_FIR_FFT_EDGE_CACHED && AUX != 0 && (row < EDGE || row >= 64 - EDGE)) // `row` is a constant after unrolling
        load_row_aux<I16, 0>(r, rsrc, lane, row);
    else
        load_row_aux<I16, AUX>(r, rsrc, lane, row);
}
__device__ __forceinline__ void buf_store(srd_t rsrc, unsigned voff, unsigned soff, cf d)
{
    u32x2_t v;
    v[0] = __float_as_uint(d.x);
    v[1] = __float_as_uint(d.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, voff, soff, IF_FIR_FFT_STORE_AUX);
}

// The filter bank's tails store per lane: every lane writes the NOUT outputs it holds to ITS channel's buffer, element k at
// pl + k STEP.  Round 4 tested `index < M` in front of every store -- a branch, an exec mask and a 64-bit address per output.  Here a
// block all of whose outputs exist (wave-uniform; every block of a call but possibly the last) stores through ONE per-lane base
// address with immediate offsets and no test; the last block keeps the per-output test.
template <int NOUT, int STEP, typename F>
__device__ __forceinline__ void store_lane_rows(cf *pl, bool full, int64_t idx0, int64_t M, F &&value)
{
    if (full)
    {
#pragma unroll
        for (int k = 0; k < NOUT; k++)
            __builtin_nontemporal_store(value(k), pl + k * STEP);
    }
    else
    {
#pragma unroll
        for (int k = 0; k < NOUT; k++)
            if (idx0 + k * STEP < M)
                __builtin_nontemporal_store(value(k), pl + k * STEP);
    }
}

// Decimations D = 4 * sub (8, 12, ..., 64) behind the decimate-by-4 tail, D = 2 * sub (6, 10, ..., 62) behind the decimate-by-2 one: the tail
// runs at the fs/F rate and every sub-th of its outputs is a real output.  The block grid starts at a kept output (the launcher
// shifts it by the call's decimation phase), so tail output number i (counted over the whole call) is kept when i is a
// multiple of sub, as output i / sub.  The block's share (obase) is divided once per block, wave-uniform, in SGPRs; each
// output then costs a multiply-shift (ceil(2^18 / sub), exact for numerators below 2^12: remainder + lane offset + step
// < 2100; checked over the whole range by tests/test_host.py).  sub = 1 keeps everything.
struct KeepEvery
{
    int64_t qU;     // floor(obase / sub), wave-uniform
    unsigned rem;   // obase mod sub, wave-uniform
    unsigned magic; // ceil(2^18 / sub)
    unsigned sub;
    // obase = blk * lout (blk < 2^31, lout <= 1920; 1, 2, 4, 8, 16: a shift): divided in 32-bit pieces, blk = bq sub + br ->
    // obase / sub = bq lout + (br lout) / sub -- a 64-bit division here costs a dozen temporaries the tails do not have
    __device__ __forceinline__ void init(int64_t blk, unsigned lout, unsigned sub_)
    {
        sub = sub_ ? sub_ : 1u;
        uint64_t q;
        if ((sub & (sub - 1u)) == 0u) // a power of two: a shift
        {
            const uint64_t ob = (uint64_t)blk * lout;
            rem = (unsigned)ob & (sub - 1u);
            q = ob >> __builtin_ctz(sub);
        }
        else
        {
            const unsigned bq = (unsigned)blk / sub, br = (unsigned)blk - bq * sub;
            const unsigned t = br * lout, q2 = t / sub;
            rem = t - q2 * sub;
            q = (uint64_t)bq * lout + q2;
        }
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q), hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
        qU = (int64_t)(((uint64_t)hi << 32) | lo);
        rem = (unsigned)__builtin_amdgcn_readfirstlane(rem);
        magic = (unsigned)__builtin_amdgcn_readfirstlane((262144u + sub - 1u) / sub);
    }
    // tail output `off` of this block (lane offset + step): its index among the call's kept outputs, or -1
    __device__ __forceinline__ int64_t index(unsigned off) const
    {
        const unsigned u = rem + off, qd = (u * magic) >> 18;
        return (u - qd * sub == 0u) ? qU + (int64_t)qd : (int64_t)-1;
    }
};

// ---- block queue (two levels): if_fir_fft_queue.h, shared with the host simulation --------------------------------------
// Global queue block (32 bytes per context): words 0, 1 = group ticket counters (launches alternate; each launch zeroes the
// other one for the launch behind it), words 2, 3 = tail ticket counters (likewise), word 4 = expired bounded waits.
struct DevQueue
{
    char *qcur;   // LDS: the current-group word (8 B, 16-byte slot) followed by the look-ahead ring
    char *qtail;  // LDS: the tail word
    char *qclaim; // LDS: the four SIMDs' tail claim counters
    unsigned int *gqueue; // this launch's global ticket counter
    unsigned int *tqueue; // this launch's tail ticket counter
    unsigned int *faultw; // bounded waits that expired (0 in a healthy launch)
    int lane;
    __device__ __forceinline__ unsigned long long *tailw() const { return reinterpret_cast<unsigned long long *>(qtail); }
    __device__ __forceinline__ unsigned tail_claim(unsigned simd)
    {
        unsigned c = 0;
        if (lane == 0)
            c = __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(qclaim) + simd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(c);
    }
    __device__ __forceinline__ unsigned long long tail_add()
    {
        unsigned long long w = 0;
        if (lane == 0)
            w = __hip_atomic_fetch_add(tailw(), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return uniform(w);
    }
    __device__ __forceinline__ unsigned long long tail_load()
    {
        return uniform(__hip_atomic_load(tailw(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    __device__ __forceinline__ void tail_store(unsigned long long v)
    {
        if (lane == 0)
            __hip_atomic_store(tailw(), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ unsigned tail_ticket()
    {
        unsigned t = 0;
        if (lane == 0)
            t = atomicAdd(tqueue, 1u);
        return __builtin_amdgcn_readfirstlane(t);
    }
    __device__ __forceinline__ unsigned long long *cur() const { return reinterpret_cast<unsigned long long *>(qcur); }
    __device__ __forceinline__ unsigned long long *ring() const { return reinterpret_cast<unsigned long long *>(qcur + 16); }
    static __device__ __forceinline__ unsigned long long uniform(unsigned long long v)
    {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    }
    __device__ __forceinline__ unsigned long long cur_add()
    {
        unsigned long long w = 0;
        if (lane == 0)
            w = __hip_atomic_fetch_add(cur(), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return uniform(w);
    }
    __device__ __forceinline__ unsigned long long cur_load()
    {
        return uniform(__hip_atomic_load(cur(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    __device__ __forceinline__ void cur_store(unsigned long long v)
    {
        if (lane == 0)
            __hip_atomic_store(cur(), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ unsigned long long ring_load(unsigned i)
    {
        return uniform(__hip_atomic_load(&ring()[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    __device__ __forceinline__ void ring_store(unsigned i, unsigned long long v)
    {
        if (lane == 0)
            __hip_atomic_store(&ring()[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ unsigned ticket()
    {
        unsigned t = 0;
        if (lane == 0)
            t = atomicAdd(gqueue, 1u);
        return __builtin_amdgcn_readfirstlane(t);
    }
    __device__ __forceinline__ void fault()
    {
        if (lane == 0)
            atomicAdd(faultw, 1u);
    }
    __device__ __forceinline__ void pause() { __builtin_amdgcn_s_sleep(2); }
    __device__ __forceinline__ unsigned wgs() const { return gridDim.x; }
};

// common tail of the small inverses: a[j], j = 4 i + low (low = mu2 of the 1024-point inverse, or the channel-in-batch of the
// 16-slot bank), k0 = 4 g + i, k1 = lane % 16:
//   X: row transposition (one round of exchange 2): element j of lane (g, k1) -> lane (g, j), slot k1; iFFT16 over k1 -> mu1
//   twiddle conj W256^(k0 mu1);  Y: element mu1 of lane (k0, low) -> lane 4 mu1 + low, slot k0;  iFFT16 over k0 -> mu0
// result: lane = 4 mu1 + low, slot mu0
__device__ __forceinline__ void inverse_tail256(cf (&a)[16], cf (&c)[16], const f2v *twe, const XAddr &xa, int lane)
{
    xst16<XROW>(xa.wx, a);
    xld16<8>(xa.rx, a);
    fft16<true>(a); // over k1 -> mu1
#pragma unroll
    for (int mu1 = 1; mu1 < 16; mu1++)
        a[mu1] = cmul_v<true>(a[mu1], twe[mu1 * 64 + lane]);
    // Y: wy + mu1 XROW = element (mu1, k0) of region `low`; ry + 8 k0
    xst16<XROW>(xa.wy, a);
    xld16<8>(xa.ry, c);
    fft16<true>(c); // over k0 -> mu0
}

// The same for the filter-bank images (round 5): the twiddle conj W256^(k0 mu1) between the two transforms sits on the INPUTS of the
// second one in (cos, tan) form -- input k0 of lane 4 mu1 + low carries conj(b)^k0, b = W256^mu1 (table twet[e * 16 + mu1], the 15
// entries of fft16_tw) -- 88 packed instructions where the 15 multiplies + the plain transform are 110.
__device__ __forceinline__ void inverse_tail256_tan(cf (&a)[16], cf (&c)[16], const f2v *twet, const XAddr &xa, int lane)
{
    xst16<XROW>(xa.wx, a);
    xld16<8>(xa.rx, a);
    fft16<true>(a); // over k1 -> mu1
    xst16<XROW>(xa.wy, a);
    xld16<8>(xa.ry, c);
    fft16_tw<true, 16>(c, twet + (lane >> 2)); // over k0 -> mu0
}

// decimate-by-4 tail of one block: the 4 spectral aliases are folded in-lane (k2 = k2' + 4j) and a 1024-point inverse
// (4 x 16 x 16, tools/fft_model.py inverse_dec4) produces y[4m'] directly: lane = 4*mu1+mu2, slot mu0 -> y_D[64*mu0+lane]
__device__ __forceinline__ void inverse_dec4(const cf (&z)[16], cf (&c)[16], const f2v *twd, const f2v *twe, const XAddr &xa,
                                             int lane)
{
    cf a[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        bfly4<true>(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3], a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
#pragma unroll
        for (int mu2 = 1; mu2 < 4; mu2++)
            a[4 * i + mu2] = cmul_v<true>(a[4 * i + mu2], twd[(i * 4 + mu2) * 64 + lane]);
    }
    inverse_tail256(a, c, twe, xa, lane);
}

// the same with the twiddles in (cos, tan) form on the inputs of the two 16-point transforms (round 4; tables tb = LDS_TWE,
// tc = LDS_TWD): 4-point inverse over k2' (plain) -> X -> iFFT16 over k1, inputs carry conj(W64^mu2)^k1 -> Y -> iFFT16 over k0,
// inputs carry conj(W1024^lane)^k0.  208 packed instructions where inverse_dec4 has 246.
__device__ __forceinline__ void inverse_dec4_tan(const cf (&z)[16], cf (&c)[16], const f2v *tb, const f2v *tc, const XAddr &xa, int lane)
{
    cf a[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
        bfly4<true>(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3], a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
    xst16<XROW>(xa.wx, a);
    xld16<8>(xa.rx, a);
    fft16_tw<true, 4>(a, tb + (lane & 3)); // over k1 -> mu1
    xst16<XROW>(xa.wy, a);
    xld16<8>(xa.ry, c);
    fft16_tw<true, 64>(c, tc + lane); // over k0 -> mu0
}


// ---- filter bank, channels at their own centres: one folded value of a channel (round 5) ------------------------
//     z = sum_n d[n] w[n] g[n GS],  n = 0 .. N - 1,  w[0] = 1, w[n] = tw[n - 1] wave-uniform (the channel's W4096^(n B), SGPRs),
// g = the lane's gathered table entries (LDS).  Round 4 wrote this as one chain `z = cmac(z, cmul_s(d, w), g[..])`, and the compiler
// produced exactly that: every table read directly in front of its use (a full LDS round trip exposed per term, `s_waitcnt
// lgkmcnt(0)` behind each ds_read), one dependent chain of 2 N packed FMAs, and the twiddles' scalar loads in the middle (scalar
// loads return out of order, so each of them drains the LDS reads in flight): SQ_WAIT_ANY 46-54 % of the wave cycles
// (profiles/r04_pmc_filter_bank.txt).  Here the N gathers are requested first, the N - 1 products d w -- which need no table -- are
// formed while they fly, and the multiply-accumulates run as NA interleaved partial sums.
// In batches of NB terms (registers: the next batch's gathers are hoisted above this batch's arithmetic by the scheduler, so
// two batches of table entries are live at a time): per batch the NB gathers are requested first, the products d w -- which need
// no table -- are formed while they fly, and the multiply-accumulates run as NA interleaved partial sums.
#ifndef IF_FIR_GM_NA
#define IF_FIR_GM_NA 2 // partial sums (4 with batches of 8 or 16 spills; 2 x 4: 240 VGPRs)
#endif
#ifndef IF_FIR_GM_NB
#define IF_FIR_GM_NB 8 // terms per batch (4: 1 % slower, profiles/r05_filter_bank_ab.txt)
#endif
template <int N, int GS, int NA, int NB>
__device__ __forceinline__ cf gather_mac(const cf (&d)[N], const cf (&tw)[N - 1], const f2v *g)
{
    static_assert(N % NB == 0 && NB % NA == 0, "whole batches, whole rounds of the partial sums");
    cf acc[NA];
#pragma unroll
    for (int b0 = 0; b0 < N; b0 += NB)
    {
        cf gq[NB], q[NB];
#pragma unroll
        for (int n = 0; n < NB; n++)
            gq[n] = g[(b0 + n) * GS];
        // (nothing crosses this point in the machine scheduler: left to itself it sinks every gather down to its use -- it
        // minimises live registers -- and the wave waits a full LDS round trip per term)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NB; n++)
            q[n] = (b0 + n == 0) ? d[0] : cmul_s<false>(d[b0 + n], tw[b0 + n - 1]);
#pragma unroll
        for (int n = 0; n < NB; n++)
            acc[n % NA] = (b0 + n < NA) ? cmul_v<false>(q[n], gq[n]) : cmac_v(acc[n % NA], q[n], gq[n]);
    }
#pragma unroll
    for (int k = NA / 2; k >= 1; k /= 2)
#pragma unroll
        for (int m = 0; m < k; m++)
            acc[m] = acc[m] + acc[m + k];
    return acc[0];
}

#ifdef IF_FIR_FFT_ODD // ================= odd decimations 3, 9, 15, ..., 63: their own compilation unit =================
// Round 4: 4096 has no odd factor to fold by, so odd decimations ran the full-rate pipeline with a selecting store
// (2528 packed instructions per 3840 input samples).  Here a block is F x 1024 input samples, F = 3 or 5: lane l of row r loads
// the F consecutive samples x[s0 + F (64 r + l) + p], p = 0..F-1 -- F phase streams x_p[m] = x[s0 + F m + p] in the load layout of
// a 1024-point transform -- and
//     y[F m] = sum_p (g_p * x_p)[m],   g_0[k] = h[F k],  g_p[d] = h[F d - p]  (p >= 1, d >= 1: x[F m - k] = x_p[m - d] for k = F d - p)
// is evaluated as  Z = sum_p FFT1024(x_p) G_p,  y = IFFT1024(Z):  F forward 1024-point transforms (the mirror image of the
// decimate-by-4 kernel's small inverse, twiddles in (cos, tan) form on the inputs), F x 16 complex MACs per lane and ONE
// inverse -- 3 x 212 + 96 + 208 = 940 packed instructions per 2688 input samples at F = 3 (tools/fft_model.py odd_block).
// The first 64 OVLR outputs of a block are dropped (OVLR = 2 or 4 rows: (T - 1 + F - 1) / F <= 64 OVLR; <= 383 / 767 taps).  Decimations F x sub
// (9, 15, 21, ...; 25, 35, 55) keep every sub-th output of this tail (KeepEvery, as behind the even tails).
// LDS image (fft_build_tables_odd): G_p [(p*16 + slot)*64 + lane] | TB [e*16 + k0] (forward middle pass, b = W256^k0) |
// TC [(i*3 + e)*64 + lane] (forward last pass, b = W1024^(k0 + 16 k1)) | TWD, TWE (the inverse's tables, as in the decimate-by-4
// image) | NCO row phasors
constexpr int ODD_LDS_G = 0;
template <int F> struct OddLds
{
    static constexpr int WBUF = XBUF; // per-wave LDS buffer: the exchange buffers of the transforms
    static constexpr int TB = F * 16 * 64 * 8, TC = TB + 2048, TWD = TC + 4 * 3 * 64 * 8, TWE = TWD + 8192, NCO = TWE + 512,
                         XB = NCO + 512, Q = XB + FFT_WAVES * WBUF, QTAIL = Q + 16 + Q_RING * 8, QCLAIM = QTAIL + 16, BYTES = QCLAIM + 16;
    static_assert(XB == fft_odd_table_floats(F) * 4, "odd table image size");
    static_assert(BYTES <= 160 * 1024, "LDS");
};

// forward 1024-point transform of reg[row] = x[64 row + lane] into z (slot 4 i + k2', lane (g, k1): X[k0 + 16 k1 + 256 k2'],
// k0 = 4 g + i): FFT16 over the rows (plain; done by the caller) -> Y^-1 -> FFT16 over mu1, inputs carry (W256^k0)^mu1 -> X^-1 ->
// 4-point DFT over mu2, inputs carry (W1024^(k0 + 16 k1))^mu2
// (v = the output of the first pass, the plain FFT16 over the rows, of the lane whose in-lane-order index lsrc = 4 mu1 + mu2 gave
// wr_off = (lsrc & 3) XREG + (lsrc >> 2) XROW: the kernel runs that pass on its coalesced registers and lets every lane deliver
// the column it happens to hold)
// (xa: the X / Y addresses of the inverse; ryi: the OPAQUE read base of Y^-1, the address formula of xa.wy)
__device__ __forceinline__ void forward_1024_tan(cf (&v)[16], cf (&z)[16], const f2v *tb, const f2v *tc, char *xb, const XAddr &xa,
                                                 const char *ryi, int lane, int wr_off)
{
    cf y[16];
    {
        // Y^-1: element k0 of lane 4 mu1 + mu2 -> lane (k0, mu2), slot mu1 (the addresses of inverse_tail256's Y, roles swapped):
        // written to xb + wr_off + 8 k0, read from ryi + mu1 XROW
        char *wr = xb + wr_off;
        const int k0 = 4 * (lane >> 4) + ((lane & 15) >> 2);
        xst16<8>(wr, v);
        xld16<XROW>(ryi, y);
        fft16_tw<false, 16>(y, tb + k0); // over mu1 -> k1
    }
    {
        // X^-1 (the row transposition is its own inverse): element k1 of lane (g, j) -> lane (g, k1), slot j = 4 i + mu2
        xst16<XROW>(xa.wx, y);
        xld16<8>(xa.rx, y);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const cf e1 = tc[(i * 3 + 0) * 64 + lane], e2 = tc[(i * 3 + 1) * 64 + lane], e3 = tc[(i * 3 + 2) * 64 + lane];
        bfly4_tw<false>(y[4 * i], y[4 * i + 1], y[4 * i + 2], y[4 * i + 3], e1, e2, e3, z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3]);
    }
}

template <int F, int OVLR, bool I16, bool NCO, bool SUB>
__global__ __launch_bounds__(512, 2) void fir_odd_kernel(const f2v *__restrict__ in_, f2v *__restrict__ out,
                                                        const f2v *__restrict__ tables, const f2v *__restrict__ hist, int HL, int64_t N,
                                                        int32_t n0, int64_t M, int64_t nblocks, unsigned int *queue, int32_t diag,
                                                        uint32_t nco_phi0, uint32_t nco_delta, uint32_t qsel, void *__restrict__ hist_out,
                                                        uint32_t sub, int64_t decn_m)
{
    using L = OddLds<F>;
    constexpr int ISZ = I16 ? 4 : 8;
    constexpr int LOUT = 1024 - 64 * OVLR, LIN = F * LOUT, OVL = F * 64 * OVLR;
    // phases whose refill with the next block's rows is issued behind the inverse instead of right after their transform (their
    // registers would otherwise be live through the other phases' transforms and the inverse: with one phase late the instantiations
    // without an NCO use 150-200 bytes of scratch)
    constexpr int LATE = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char *in = reinterpret_cast<const char *>(in_);
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        const f4v_t *src = reinterpret_cast<const f4v_t *>(tables);
        f4v_t *dst = reinterpret_cast<f4v_t *>(smem);
        constexpr int NV = L::XB / 16, NK = (NV + 511) / 512;
        f4v_t tv[NK];
#pragma unroll
        for (int k = 0; k < NK; k++)
            if ((int)threadIdx.x + 512 * k < NV)
                tv[k] = src[threadIdx.x + 512 * k];
#pragma unroll
        for (int k = 0; k < NK; k++)
            if ((int)threadIdx.x + 512 * k < NV)
                dst[threadIdx.x + 512 * k] = tv[k];
        if (threadIdx.x < Q_RING)
            reinterpret_cast<unsigned long long *>(smem + L::Q + 16)[threadIdx.x] = queue_ring_init(threadIdx.x, blockIdx.x, gridDim.x);
        if (threadIdx.x < 4)
            reinterpret_cast<unsigned int *>(smem + L::QCLAIM)[threadIdx.x] = 0u;
        if (threadIdx.x == 0)
        {
            *reinterpret_cast<unsigned long long *>(smem + L::Q) = queue_cur_init(blockIdx.x, gridDim.x, false);
            *reinterpret_cast<unsigned long long *>(smem + L::QTAIL) = 0ull;
            if (blockIdx.x == 0)
            {
                queue[qsel ^ 1u] = 0u;
                queue[2u + (qsel ^ 1u)] = 0u;
            }
        }
    }
    __syncthreads();
    DevQueue dq{smem + L::Q, smem + L::QTAIL, smem + L::QCLAIM, queue + qsel, queue + 2 + qsel, queue + 4, lane};
    const unsigned simd = (unsigned)wid & 3u;
    // the next call's history (as in fir_fft_kernel): the last HL samples of (history || input), one wave
    if (hist_out && blockIdx.x == 0 && wid == 0)
    {
        const int64_t keep = (int64_t)HL;
        for (int64_t i = lane; i < keep; i += 64)
        {
            const int64_t gi = N - keep + i, hi = keep + gi;
            if constexpr (I16)
            {
                const int *src = reinterpret_cast<const int *>(in_), *hsrc = reinterpret_cast<const int *>(hist);
                reinterpret_cast<int *>(hist_out)[i] = gi >= 0 ? src[gi] : (hi >= 0 ? hsrc[hi] : 0);
            }
            else
                reinterpret_cast<f2v *>(hist_out)[i] = gi >= 0 ? in_[gi] : (hi >= 0 ? hist[hi] : (f2v){0.f, 0.f});
        }
    }
    const f2v *gtab = reinterpret_cast<const f2v *>(smem + ODD_LDS_G);
    const f2v *tb = reinterpret_cast<const f2v *>(smem + L::TB);
    const f2v *tc = reinterpret_cast<const f2v *>(smem + L::TC);
    const f2v *twd = reinterpret_cast<const f2v *>(smem + L::TWD);
    const f2v *twe = reinterpret_cast<const f2v *>(smem + L::TWE);
    const f2v *ncob = reinterpret_cast<const f2v *>(smem + L::NCO);
    (void)ncob;
    char *xb = smem + L::XB + wid * L::WBUF;
    const XAddr xa = xaddr_xy(xb, lane);
    const char *ryi = lds_opaque(xa.wy); // Y^-1 of the forward transforms reads where the inverse's Y writes
    // x[p][row]: piece p of row r = the 64 samples F 64 r + 64 p + lane of the block, 512 contiguous bytes per load instruction (the
    // in-lane order -- sample F (64 row + lane) + p -- only in the first block of a call).  The first form of this kernel loaded
    // the F samples of a lane directly (8 bytes per lane, 24 apart): every instruction then touched all 12 lines of a row, three
    // times the address work of the texture unit, and the kernel ran 1.16 ms where the selecting store takes 0.76
    // (profiles/r04_odd_decimation.txt).  int16 input: the raw pair sits in .x until it is used.
    cf x[F][16];
    auto load_phase = [&](srd_t srd, int p) {
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const unsigned vo = (unsigned)lane * ISZ, so = (unsigned)((r * F + p) * 64 * ISZ);
            if constexpr (I16)
                x[p][r].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd, vo, so, IF_FIR_FFT_LOAD_AUX(0)));
            else
                x[p][r] = buf_load<IF_FIR_FFT_LOAD_AUX(0)>(srd, vo, so);
        }
    };
    // coalesced pieces -> phase streams WITHOUT a transposition of their own (the second form of this kernel had one, through LDS:
    // 96 LDS instructions and 4 round trips a block).  Register column j of lane l holds row-sample 64 j + l = F l' + p' of every
    // row, i.e. the column of in-lane-order lane l' = (64 j + l) / F of phase p' = (j + l) mod F (64 = 1 mod 3): the first pass
    // of the 1024-point transform -- the FFT16 over the rows -- runs on the columns as they are; then every lane rotates its F
    // columns by l mod F so that register column p holds phase p, and delivers it into the first transposition as lane
    // l'_p = (64 ((p - l) mod F) + l) / F would have.
    static_assert(F == 3, "column rotation written for three phases");
    const int rot = lane % 3;
    int wr_ph[F];
#pragma unroll
    for (int p = 0; p < F; p++)
    {
        const int lsrc = (64 * ((p - rot + 3) % 3) + lane) / 3;
        wr_ph[p] = (lsrc & 3) * XREG + (lsrc >> 2) * XROW;
    }
    const int wr_std = (lane & 3) * XREG + (lane >> 2) * XROW;
    int64_t blk = queue_take(dq, simd, nblocks, nblocks);
    bool loaded = false;
    const unsigned voff = (unsigned)lane * 8u;
    while (blk < nblocks)
    {
        const int64_t s0 = blk * LIN - OVL + n0;
        bool inlane = false;
        if (!loaded && !(diag & 1))
        {
            if (s0 >= 0)
            {
                const srd_t srd = make_srd(in + s0 * ISZ, (N - s0) * ISZ);
#pragma unroll
                for (int p = 0; p < F; p++)
                    load_phase(srd, p);
            }
            else
            {
                // first block of a call: negative stream indices come from the history (or are zero)
                const srd_t srd_in = make_srd(in, N * ISZ);
                const srd_t srd_h = make_srd(hist, (int64_t)HL * ISZ);
#pragma unroll
                for (int p = 0; p < F; p++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                    {
                        const int64_t gidx = s0 + (int64_t)F * (64 * r + lane) + p, hidx = gidx + HL;
                        const unsigned oi = (gidx >= 0) ? (unsigned)gidx * (unsigned)ISZ : 0x80000000u;
                        const unsigned oh = (gidx < 0 && hidx >= 0) ? (unsigned)hidx * (unsigned)ISZ : 0x80000000u;
                        if constexpr (I16)
                            x[p][r].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd_in, oi, 0, 0) |
                                                        __builtin_amdgcn_raw_buffer_load_b32(srd_h, oh, 0, 0));
                        else
                            x[p][r] = buf_load(srd_in, oi, 0) + buf_load(srd_h, oh, 0);
                    }
                inlane = true; // (this block was fetched in the lanes' own order: no transposition)
            }
        }
        // first pass of the three 1024-point transforms on the register columns as loaded, then the column rotation
#pragma unroll
        for (int j = 0; j < F; j++)
        {
            cf v[16];
#pragma unroll
            for (int r = 0; r < 16; r++)
                v[r] = I16 ? cvt_i16(__float_as_uint(x[j][r].x)) : x[j][r];
            fft16<false>(v);
#pragma unroll
            for (int r = 0; r < 16; r++)
                x[j][r] = v[r];
        }
        {
            const bool r1 = !inlane && rot == 1, r2 = !inlane && rot == 2;
#pragma unroll
            for (int k0 = 0; k0 < 16; k0++)
            {
                const cf a0 = x[0][k0], a1 = x[1][k0], a2 = x[2][k0];
                x[0][k0] = r1 ? a2 : r2 ? a1 : a0;
                x[1][k0] = r1 ? a0 : r2 ? a2 : a1;
                x[2][k0] = r1 ? a1 : r2 ? a0 : a2;
            }
        }
        int64_t blk_next = nblocks;
        bool next_fast = false;
        srd_t nsrd = make_srd(in, 0);
        cf zacc[16];
#pragma unroll
        for (int p = 0; p < F; p++)
        {
            cf v[16], z[16];
#pragma unroll
            for (int r = 0; r < 16; r++)
                v[r] = x[p][r];
            forward_1024_tan(v, z, tb, tc, xb, xa, ryi, lane, inlane ? wr_std : wr_ph[p]);
#pragma unroll
            for (int sidx = 0; sidx < 16; sidx++)
            {
                const cf gw = gtab[(p * 16 + sidx) * 64 + lane];
                zacc[sidx] = p == 0 ? cmul_v<false>(z[sidx], gw) : cmac_v(zacc[sidx], z[sidx], gw);
            }
            if (p == 0)
            {
                // the next block is taken here: this block's rows have all landed, none of the next one's is in flight
                blk_next = queue_take(dq, simd, nblocks, nblocks);
                const int64_t s0n = blk_next * LIN - OVL + n0;
                next_fast = (blk_next < nblocks) && (s0n >= 0) && !(diag & 1);
                nsrd = make_srd(in + (next_fast ? s0n : 0) * ISZ, next_fast ? (N - s0n) * ISZ : 0);
            }
            // this phase's 16 registers are dead: refill them with the next block's (the last LATE phases behind the inverse)
            if (next_fast && p < F - LATE)
                load_phase(nsrd, p);
        }
        cf c[16];
        inverse_dec4_tan(zacc, c, twe, twd, xa, lane);
        if (next_fast)
        {
#pragma unroll
            for (int p = F - LATE; p < F; p++)
                load_phase(nsrd, p);
        }
        const int64_t obase = blk * LOUT;
        // SPEC 3.2: output m = obase + 64 (mu0 - OVLR) + lane of the fs/F-rate tail is rotated by phasor(phi0 + delta m) = A(lane) B(row)
        cf a_lane = {1.0f, 0.0f};
        if constexpr (NCO)
        {
            const float2 pa = nco_phasor(nco_phi0 + nco_delta * ((uint32_t)obase + (uint32_t)lane));
            a_lane = (cf){pa.x, pa.y};
        }
        (void)a_lane;
        if constexpr (SUB)
        {
            KeepEvery ke;
            ke.init(blk, (unsigned)LOUT, sub);
            const int64_t qb = ke.qU + (ke.rem ? 1 : 0);
            const srd_t dsrd = make_srd(out + qb, (diag & 2) ? 0 : (decn_m - qb) * 8);
            const int lim = (int)((M - obase) < 65536 ? (M - obase) : 65536);
            int off = lane;
#pragma unroll
            for (int mu0 = OVLR; mu0 < 16; mu0++)
            {
                // (one running offset, made opaque: otherwise the store offsets are all computed ahead of the inverse and spill)
                asm volatile("" : "+v"(off));
                const int64_t kept = ke.index((unsigned)off);
                const unsigned so = (kept >= 0 && off < lim) ? (unsigned)(kept - qb) * 8u : 0xffffffffu;
                cf v = c[mu0];
                if constexpr (NCO)
                    v = cmul_v<false>(v, cmul_v<false>(a_lane, ncob[mu0 - OVLR]));
                buf_store(dsrd, so, 0, v);
                off += 64;
            }
        }
        else
        {
            const srd_t osrd = make_srd(out + obase, (diag & 2) ? 0 : (M - obase) * 8);
#pragma unroll
            for (int mu0 = OVLR; mu0 < 16; mu0++)
            {
                cf v = c[mu0];
                if constexpr (NCO)
                    v = cmul_v<false>(v, cmul_v<false>(a_lane, ncob[mu0 - OVLR]));
                buf_store(osrd, voff, (mu0 - OVLR) * 512, v);
            }
        }
        loaded = next_fast;
        blk = blk_next;
    }
}

template <int F, int OVLR, bool I16, bool NCO, bool SUB>
static hipError_t launch_odd_t(const LaunchArgs &a, int sub)
{
    auto kern = fir_odd_kernel<F, OVLR, I16, NCO, SUB>;
    constexpr int LOUT = 1024 - 64 * OVLR;
    static DeviceSetup setup;
    int ncus = 0;
    {
        const hipError_t e = device_setup(setup, a.device, reinterpret_cast<const void *>(kern), OddLds<F>::BYTES, &ncus);
        if (e != hipSuccess)
            return e;
    }
    const int64_t m_rate = (a.M - 1) * (int64_t)sub + 1; // outputs of the fs/F-rate tail that the call's outputs need
    const int64_t nblocks = a.M > 0 ? (m_rate + LOUT - 1) / LOUT : 0;
    if (nblocks <= 0)
        return hipSuccess;
    const int64_t wgs_max = (a.grid_limit > 0 && a.grid_limit < ncus) ? a.grid_limit : ncus;
    FftSchedule sch;
    fft_schedule(nblocks, wgs_max, sch);
    uint32_t qsel = 0;
    if (a.queue_base && a.queue_valid && *a.queue_valid)
        qsel = *a.queue_base & 1u;
    else
    {
        hipError_t e = hipMemsetAsync(a.queue, 0, 16, a.stream);
        if (e != hipSuccess)
            return e;
    }
    if (a.queue_base && a.queue_valid)
    {
        *a.queue_base = qsel ^ 1u;
        *a.queue_valid = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)sch.wgs), dim3(512), OddLds<F>::BYTES, a.stream, reinterpret_cast<const f2v *>(a.in),
                       reinterpret_cast<f2v *>(a.out), reinterpret_cast<const f2v *>(a.fft_tables),
                       reinterpret_cast<const f2v *>(a.hist_full), a.hist_len, a.N, (int32_t)a.n0, m_rate, nblocks, (unsigned int *)a.queue,
                       (int32_t)a.diag, nco_phi0(a), 0u - a.nco_word * (uint32_t)F, qsel, a.hist_out, (uint32_t)sub, a.M);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess && a.queue_valid)
        *a.queue_valid = false;
    return le;
}

hipError_t launch_fft_odd(const LaunchArgs &a)
{
    int F = 1, sub = 1, ovlr = 0;
    if (!fft_odd_tail(a.T, a.D, &F, &sub, &ovlr) || a.chan || !a.fft_tables)
        return hipErrorInvalidConfiguration;
    const int key = (a.in_i16 ? 2 : 0) | (a.nco_word ? 1 : 0);
#define IF_FIR_ODD_SWITCH(F_, R_)                                                                                         \
    if (F == F_ && ovlr == R_)                                                                                            \
    {                                                                                                                     \
        if (sub == 1)                                                                                                     \
            switch (key)                                                                                                  \
            {                                                                                                             \
            case 0: return launch_odd_t<F_, R_, false, false, false>(a, sub);                                             \
            case 1: return launch_odd_t<F_, R_, false, true, false>(a, sub);                                              \
            case 2: return launch_odd_t<F_, R_, true, false, false>(a, sub);                                              \
            default: return launch_odd_t<F_, R_, true, true, false>(a, sub);                                              \
            }                                                                                                             \
        switch (key)                                                                                                      \
        {                                                                                                                 \
        case 0: return launch_odd_t<F_, R_, false, false, true>(a, sub);                                                  \
        case 1: return launch_odd_t<F_, R_, false, true, true>(a, sub);                                                   \
        case 2: return launch_odd_t<F_, R_, true, false, true>(a, sub);                                                   \
        default: return launch_odd_t<F_, R_, true, true, true>(a, sub);                                                   \
        }                                                                                                                 \
    }
    IF_FIR_ODD_SWITCH(3, 2)
    IF_FIR_ODD_SWITCH(3, 4)
#undef IF_FIR_ODD_SWITCH
    return hipErrorInvalidConfiguration;
}
#endif // IF_FIR_FFT_ODD

#ifdef IF_FIR_FFT_ROWS // ================= kernel + launcher: the per-overlap-length compilation units =================
// The tail of a block after the forward transform, by CHAN (DEC4 = any decimating tail):
//    0  full rate (DEC4 = false; DECN: selecting store), or the decimate-by-4 tail       1  decimate-by-4 tail keeping every sub-th output
//    2  decimate-by-2 tail                                                                3  the same keeping every sub-th output
//    4  filter bank at decimation 4, channels on the fs/16 slot grid (per channel)        5  the same, every channel at its own centre bin
//    6  tail 5 keeping every sub-th output (decimation 12, 20, 28, ...; tails 8-general and 17 do that inside, by a wave-uniform branch)
//    8  filter bank at decimation 8 per channel: slot grid (NCO = false) / any centre bin or a common offset (NCO = true)
//    9  filter bank at decimation 8, all slots of one parity from two 8-point transforms per group
//   16  filter bank at decimation 16, all 16 slots from one 16-point transform per group (NCO: a common offset)
//   17  filter bank at decimation 16 per channel, every channel at its own centre bin
template <int OVL_ROWS, bool DEC4, bool I16, bool NCO, int CHAN, bool DECN, bool ACC>
__global__ __launch_bounds__(512, 2) void fir_fft_kernel(const f2v *__restrict__ in_, f2v *__restrict__ out,
                                                        const f2v *__restrict__ tables, const f2v *__restrict__ hist,
                                                        int HL, int64_t N, int32_t n0, int64_t M, int64_t nblocks,
                                                        int64_t nblocks_main, unsigned int *queue, unsigned long long *dbg, int32_t diag,
                                                        uint32_t nco_phi0, uint32_t nco_delta, chan_arg_t<CHAN> chan,
                                                        uint32_t qsel, void *__restrict__ hist_out,
                                                        int32_t decn, int32_t decn_n0, int64_t decn_m, int32_t in_shift)
{
    // ACC (filters of 3074..4096 taps, two partitions of <= 2048 taps): this launch filters the input DELAYED by in_shift
    // samples with the second partition's table and adds its result to what the first launch stored
    // (round 3: also behind the single-channel decimating tails, CHAN 0..3, so that two-partition filters decimate in the
    // frequency domain like shorter ones)
    static_assert(!ACC || (CHAN <= 3 && OVL_ROWS == 32), "accumulating store: single-channel pipelines, 32 overlap rows");
    static_assert(!DECN || (!DEC4 && !CHAN), "general decimation = the full-rate pipeline with a selecting store");
    // CHAN names the decimating tail beyond the plain decimate-by-4 one: 2 = single channel, decimation 2 (frequency-domain
    // fold + 2048-point inverse); 4 / 8 / 16 = the filter bank at that decimation
    // (1 = the decimate-by-4 tail keeping every sub-th output: decimation 8, 12, ..., 64; 3 = the decimate-by-2 tail doing the same:
    // decimation 6, 10, ..., 62)
    // (9, round 4 = the bank at decimation 8 in its all-slots form: the eight slots of ONE parity from two 8-point transforms per group)
    static_assert(CHAN == 0 || ((CHAN == 1 || CHAN == 2 || CHAN == 3 || CHAN == 4 || CHAN == 5 || CHAN == 6 || CHAN == 8 || CHAN == 9 || CHAN == 16 || CHAN == 17) && DEC4),
                  "decimating tails: 1, 2, 3, or the bank at 4 (4: slots, 5: any centre), 8 (8: per channel, 9: all slots of a parity), 16 (16: all slots, 17: per channel)");
    static_assert((CHAN != 5 && CHAN != 6) || !NCO, "channels at their own centres: no common offset on top");
    static_assert(CHAN != 17 || !NCO, "channels at their own centres: no common offset on top");
    static_assert(CHAN != 4 || !NCO, "the decimate-by-4 bank takes no NCO (a single channel with an NCO is the DEC4 kernel)");
    // 2 overlap rows (<= 129 taps, round 4): the full-rate pipeline only -- the decimating tails drop whole 64-output rows of the
    // fs/F-rate block (OVL_ROWS / 4, / 2, ...), which 128 samples are not
    static_assert(OVL_ROWS >= 4 || (!DEC4 && !ACC), "2 overlap rows: full-rate pipeline (D = 1, odd D) only");
    // diag (development only, results are wrong when set): 1 = skip the global loads, 2 = skip the global stores
    // decimate-by-4 kernels (single channel incl. the multiples of 4, and the bank at decimation 4): twiddles in (cos, tan) form
    // on the inputs of passes 2 and 3 and of the small inverse (round 4); every other tail keeps round 3's form and tables
    constexpr bool TAN = IF_FIR_FFT_TAN && DEC4 && (CHAN == 0 || CHAN == 1 || CHAN == 4 || CHAN == 5 || CHAN == 6 || CHAN == 8 || CHAN == 9 || CHAN == 16 || CHAN == 17); // (8, 9, 16, 17: the banks' own images)
    // the full-rate pipeline the same way, forward and inverse (the inverse's twiddles already sat on the inputs of its passes)
    constexpr bool TANF = IF_FIR_FFT_TAN && !DEC4;
    constexpr int OVL = 64 * OVL_ROWS;
    constexpr int ISZ = I16 ? 4 : 8;       // bytes per input sample
    const char *in = reinterpret_cast<const char *>(in_);
    constexpr int L = FFT_N - OVL;         // new input samples per block
    constexpr int LOUT = (CHAN == 16 || CHAN == 17) ? L / 16 : (CHAN == 8 || CHAN == 9) ? L / 8 : (CHAN == 2 || CHAN == 3) ? L / 2 : DEC4 ? L / 4 : L; // outputs per block (per channel)
    constexpr int EARLY_GROUPS = IF_FIR_FFT_EARLY_GROUPS; // dec4: batches of next-block loads issued during pass 3
    constexpr int LAUX = IF_FIR_FFT_LOAD_AUX(OVL_ROWS); // cache policy of the row loads
    constexpr int EDGE_MIN = DEC4 ? IF_FIR_FFT_EDGE_MIN_DEC : IF_FIR_FFT_EDGE_MIN_FULL;
    constexpr int EDGE = OVL_ROWS < EDGE_MIN ? EDGE_MIN : OVL_ROWS; // first / last rows of a block loaded with the default policy
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // diag 512 (development: the fault path's test): the waves of workgroup 0 behave like waves whose bounded wait has expired --
    // they count a fault and leave (the whole workgroup, ahead of its barrier), their blocks stay unwritten -- and the caller
    // must be told (if_fir_synchronize)
    if ((diag & 512) && blockIdx.x == 0)
    {
        if (lane == 0)
            atomicAdd(queue + 4, 1u);
        return;
    }
    cf r[64];
    // CHAN 9 with both slot parities in ONE launch (chan.sub bit 1; round 4): the queue hands out VIRTUAL blocks 2 b + parity -- block b of
    // the stream is transformed twice, by neighbouring waves of a workgroup, and the second read of its rows is served by L2
    const bool both = (CHAN == 9) && ((chan.sub & 2u) != 0u);
    auto rb = [&](int64_t b) -> int64_t { return (CHAN == 9 && both) ? (b >> 1) : b; }; // virtual block -> block of the stream
    bool loaded = false; // the rows of `blk` are already in flight (issued by the prologue or the previous iteration's epilogue)
    // ---- first block: static (wave w of workgroup b takes block w of global group b), and its rows are requested BEFORE
    //      the table copy below, so that the two transfers overlap at the head of the launch
    const bool plain_start = IF_FIR_FFT_LOADS_FIRST && !(diag & (32 | 64));
    int64_t blk = 0;
    if (plain_start)
    {
        blk = (int64_t)blockIdx.x * FFT_WAVES + wid; // slot wid of local group 0 = global group blockIdx.x
        const int64_t s0 = rb(blk) * L - OVL + n0 - in_shift;
        if (blk < nblocks && s0 >= 0 && !(diag & 1))
        {
            const srd_t srd = make_srd(in + s0 * ISZ, (N - s0) * ISZ);
#pragma unroll
            for (int rho = 0; rho < 4; rho++)
#pragma unroll
                for (int j = 0; j < 16; j++)
                    load_row<I16, LAUX, EDGE>(r, srd, lane, 4 * j + rho);
            loaded = true;
        }
    }
    // ---- tables: global -> LDS (once per workgroup) ------------------------------------------------------------
    {
        const f4v_t *src = reinterpret_cast<const f4v_t *>(tables);
        f4v_t *dst = reinterpret_cast<f4v_t *>(smem);
#if IF_FIR_FFT_TABLE_COPY_UNROLLED
        // all 11 loads of a thread in flight before the first LDS write: as a plain loop the compiler waits for each load
        // before the next one (11 memory round trips, ~8 us at the head of every launch with nothing else running on the CU)
        constexpr int NV = LDS_XB / 16, NK = (NV + 511) / 512;
        f4v_t tv[NK];
#pragma unroll
        for (int k = 0; k < NK; k++)
            if ((int)threadIdx.x + 512 * k < NV)
                tv[k] = src[threadIdx.x + 512 * k];
#pragma unroll
        for (int k = 0; k < NK; k++)
            if ((int)threadIdx.x + 512 * k < NV)
                dst[threadIdx.x + 512 * k] = tv[k];
#else
        for (int i = threadIdx.x; i < LDS_XB / 16; i += 512)
            dst[i] = src[i];
#endif
        if constexpr (CHAN == 16 || CHAN == 9)
        {
            if (threadIdx.x < 16)
                reinterpret_cast<float2 **>(smem + LDS_QPTR)[threadIdx.x] = chan.out[threadIdx.x];
        }
        if constexpr (CHAN == 8 || CHAN == 17 || CHAN == 5 || CHAN == 6)
        {
            // per channel, the phasors of output rows 0..15 of a block: row k is 32 outputs (decimation 16: 16, decimation 4: 64) = 256
            // input samples behind row 0
            if (threadIdx.x < 16u * chan.count)
            {
                const float2 w = nco_phasor(0u - chan.pword[threadIdx.x >> 4] * 256u * (threadIdx.x & 15u));
                reinterpret_cast<cf *>(smem + LDS_ROWT)[threadIdx.x] = (cf){w.x, w.y};
            }
        }
        // block queue (if_fir_fft_queue.h): the current-group word and the look-ahead ring
        if (threadIdx.x < Q_RING)
            reinterpret_cast<unsigned long long *>(smem + LDS_Q + 16)[threadIdx.x] = queue_ring_init(threadIdx.x, blockIdx.x, gridDim.x);
        if (threadIdx.x < 4)
            reinterpret_cast<unsigned int *>(smem + LDS_QCLAIM)[threadIdx.x] = 0u;
        if (threadIdx.x == 0)
        {
            *reinterpret_cast<unsigned long long *>(smem + LDS_Q) = queue_cur_init(blockIdx.x, gridDim.x, plain_start);
            *reinterpret_cast<unsigned long long *>(smem + LDS_QTAIL) = 0ull;
            // the other global counters are the next launch's: zero them here (this launch never touches them)
            if (blockIdx.x == 0)
            {
                queue[qsel ^ 1u] = 0u;
                queue[2u + (qsel ^ 1u)] = 0u;
            }
        }
    }
    __syncthreads();
    DevQueue dq{smem + LDS_Q, smem + LDS_QTAIL, smem + LDS_QCLAIM, queue + qsel, queue + 2 + qsel, queue + 4, lane};
    const unsigned simd = (unsigned)wid & 3u; // waves w and w + 4 of a workgroup share a SIMD
    if (plain_start && wid == 0)
        queue_start(dq); // the fetch the (static) slot 0 of local group 0 owes
    // streaming state: the history of the NEXT call = the last HL samples of (history || input) (HL = the block overlap,
    // >= T-1: the first block of a call then sees the very samples an interior block sees), written to the other
    // ping-pong buffer by one wave (everything it reads is read-only in this launch); spares a launch per call
    if (hist_out && blockIdx.x == 0 && wid == 0)
    {
        const int64_t keep = (int64_t)HL;
        for (int64_t i = lane; i < keep; i += 64)
        {
            const int64_t gi = N - keep + i, hi = keep + gi;
            if constexpr (I16)
            {
                const int *src = reinterpret_cast<const int *>(in_), *hsrc = reinterpret_cast<const int *>(hist);
                reinterpret_cast<int *>(hist_out)[i] = gi >= 0 ? src[gi] : (hi >= 0 ? hsrc[hi] : 0);
            }
            else
                reinterpret_cast<f2v *>(hist_out)[i] = gi >= 0 ? in_[gi] : (hi >= 0 ? hist[hi] : (f2v){0.f, 0.f});
        }
    }
    const f2v *tw1 = reinterpret_cast<const f2v *>(smem + LDS_TW1);
    const f2v *hp = reinterpret_cast<const f2v *>(smem + LDS_HP);
    const f2v *tw2 = reinterpret_cast<const f2v *>(smem + LDS_TW2);
    const f2v *twd = reinterpret_cast<const f2v *>(smem + LDS_TWD);
    const f2v *twe = reinterpret_cast<const f2v *>(smem + LDS_TWE);
    const f2v *twf = reinterpret_cast<const f2v *>(smem + LDS_TWF);  // decimate-by-2 inverse: W2048^(16 k1 + k0)
    (void)twf;
    const f2v *ncob = reinterpret_cast<const f2v *>(smem + LDS_NCO); // NCO: phasor of output row r of a block
    (void)ncob;
    const f2v *pht = reinterpret_cast<const f2v *>(smem + LDS_PH);   // phasor tables (the (cos, tan) images: lds_phasor)
    (void)pht;
    auto phasor = [&](uint32_t ph) -> float2 { // exp(+j 2 pi ph / 2^32): from the tables where the image has them
        if constexpr (TAN || TANF)
        {
            const cf w = lds_phasor(pht, ph);
            return make_float2(w.x, w.y);
        }
        else
            return nco_phasor(ph);
    };
    char *xb = smem + LDS_XB + wid * XBUF;
    const XAddr xa = DEC4 ? xaddr_xy(xb, lane) : xaddr_x(xb, lane); // the lane's exchange addresses (read bases opaque)
    (void)twd;
    (void)twe;
    // full-rate pipeline: the shared table T and this lane's three positions in it (tsw)
    const f2v *tt = reinterpret_cast<const f2v *>(smem + LDS_TT);
    const unsigned t_fwd3 = tsw(4u * ((unsigned)lane >> 4) + 16u * ((unsigned)lane & 15u)); // + i: xor (bits 0, 1 of the position)
    const unsigned t_inv1 = tsw((unsigned)lane), t_inv1s = tsw(4u * (unsigned)lane);
    (void)tt; (void)t_fwd3; (void)t_inv1; (void)t_inv1s;

    // diagnostics (only with a debug buffer): phase stamps of the first 32 iterations of a few waves
    int dbg_it = 0;
    const bool dbg_on = dbg && ((blockIdx.x & 63) == 0) && (wid < 2);
    (void)dbg_it;
    (void)dbg_on;
#ifdef IF_FIR_FFT_STAMPS
#define FFT_STAMP(slot)                                                                                          \
    do                                                                                                           \
    {                                                                                                            \
        if (dbg_on && dbg_it < 32 && lane == 0)                                                                  \
        {                                                                                                        \
            dbg[(((blockIdx.x >> 6) * 2 + wid) * 32 + dbg_it) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();   \
            dbg[4096 + (((blockIdx.x >> 6) * 2 + wid) * 32 + dbg_it) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
        }                                                                                                        \
    } while (0)
#else
#define FFT_STAMP(slot) (void)0
#endif
    // whole-launch stamps per wave (only with a debug buffer: if_fir_debug_stamps): realtime (100 MHz) and shader clock
    unsigned long long st_r0 = 0, st_c0 = 0;
    if (dbg)
    {
        st_r0 = __builtin_amdgcn_s_memrealtime();
        st_c0 = __builtin_amdgcn_s_memtime();
    }
    // Work distribution: if_fir_fft_queue.h
    const int32_t waves_total = (int32_t)gridDim.x * FFT_WAVES;
    // diag 64 (development, results stay correct): waves 4-7 of every workgroup leave at once = one wave per SIMD
    // (occupancy experiment; the queue hands their share to the others)
    if (!plain_start)
        blk = (diag & 32) ? 0 : ((diag & 64) && wid >= FFT_WAVES / 2) ? nblocks : queue_take(dq, simd, nblocks_main, nblocks);
    // diag 32 (development, results stay correct): static wave-interleaved blocks, no queue: block = it * waves + wave
    const bool static_map = (diag & 32) != 0;
    const int act_waves = (diag & 64) ? FFT_WAVES / 2 : FFT_WAVES;
    const int64_t static_stride = (int64_t)(waves_total / FFT_WAVES) * act_waves;
    if (static_map)
    {
        blk = (wid < act_waves) ? (int64_t)blockIdx.x * act_waves + wid : nblocks;
    }
    const unsigned voff = (unsigned)lane * 8u;
    while (blk < nblocks)
    {
        FFT_STAMP(0);
        const int64_t s0 = rb(blk) * L - OVL + n0 - in_shift; // stream index of the block's first sample (n0: decimation phase)
        if (!loaded && !(diag & 1))
        {
            if (s0 >= 0)
            {
                // rows beyond the end of the input read 0 through the descriptor's bounds check
                const srd_t srd = make_srd(in + s0 * ISZ, (N - s0) * ISZ);
#pragma unroll
                for (int rho = 0; rho < 4; rho++)
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        load_row<I16, LAUX, EDGE>(r, srd, lane, 4 * j + rho);
            }
            else
            {
                // first block of a call: negative stream indices come from the history (or are zero); one of the two
                // loads of every element is out of range and returns 0
                // (an int16 stream keeps its history as raw int16 pairs too, so the block stays raw: OR of the two loads)
                const srd_t srd_in = make_srd(in, N * ISZ);
                const srd_t srd_h = make_srd(hist, (int64_t)HL * ISZ);
#pragma unroll
                for (int row = 0; row < 64; row++)
                {
                    const int64_t gidx = s0 + row * 64 + lane;
                    const int64_t hidx = gidx + HL;
                    const unsigned oi = (gidx >= 0) ? (unsigned)gidx * (unsigned)ISZ : 0x80000000u;
                    const unsigned oh = (gidx < 0 && hidx >= 0) ? (unsigned)hidx * (unsigned)ISZ : 0x80000000u;
                    if constexpr (I16)
                        r[row].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd_in, oi, 0, 0) |
                                                   __builtin_amdgcn_raw_buffer_load_b32(srd_h, oh, 0, 0));
                    else
                        r[row] = buf_load(srd_in, oi, 0) + buf_load(srd_h, oh, 0);
                }
            }
        }

        FFT_STAMP(1);
        // ---- forward ------------------------------------------------------------------------------------------
#pragma unroll
        for (int rho = 0; rho < 4; rho++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
                t[j] = I16 ? cvt_i16(__float_as_uint(r[4 * j + rho].x)) : r[4 * j + rho];
            fft16<false>(t);
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                if (j == 0 || TAN || TANF) // (the twiddle W4096^((lane + 64 rho) k0) is applied on the inputs of passes 2 and 3)
                    r[4 * j + rho] = t[j];
                else
                {
#ifdef IF_FIR_DIAG_NO_TW // (timing study builds only: twiddles from registers instead of LDS, results wrong)
                    const f2v w = {0.6f, 0.8f};
#else
                    const f2v w = tw1[(rho * 16 + j) * 64 + lane];
#endif
                    r[4 * j + rho] = cmul_v<false>(t[j], w);
                }
            }
        }
        FFT_STAMP(2);
#ifndef IF_FIR_DIAG_NO_X1 // (timing study builds only: results are wrong without the exchange)
        exchange1_fwd(r);
#endif
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            cf t[16];
#pragma unroll
            for (int j = 0; j < 16; j++)
                t[j] = r[phys(i, j)];
            if constexpr (TAN || TANF)
            {
                // inputs carry W256^(n1 k0), k0 = 4 (lane / 16) + i: the part of pass 1's twiddle that depends on n1; the rest,
                // W4096^(n2 k0), joins this pass's own W256^(n2 k1) on the inputs of pass 3
                fft16_tw<false, 4>(t, tw2 + i * 60 + (lane >> 4));
#pragma unroll
                for (int j = 0; j < 16; j++)
                    r[phys(i, j)] = t[j];
            }
            else
            {
            fft16<false>(t);
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                if (j == 0)
                    r[phys(i, j)] = t[j];
                else
                {
#ifdef IF_FIR_DIAG_NO_TW
                    const f2v w = {0.6f, 0.8f};
#else
                    const f2v w = tw2[j * 16 + (lane & 15)];
#endif
                    r[phys(i, j)] = cmul_v<false>(t[j], w);
                }
            }
            }
        }
        FFT_STAMP(3);
        exchange2(r, xa);
        FFT_STAMP(4);
        int64_t blk_next = blk + 1;
        if (static_map)
        {
            blk_next = blk + static_stride;
        }
        else
        {
            // taken here: the block's own rows have all landed and the next block's are not issued yet, so the wait
            // behind the (rare) global atomic inside drains nothing
            // (Round 4 tried an "end game": during the launch's last one or two groups per workgroup the take was deferred until the
            // current block was stored, so that no wave holds one and a half blocks while another leaves empty-handed.  Measured
            // -0.4 % on 2^26 samples with one buffer, +0.1..0.5 % on rotating buffers and on 2^28-sample launches: removed,
            // profiles/r04_end_game.txt.)
            blk_next = queue_take(dq, simd, nblocks_main, nblocks);
        }
        const int64_t s0n = rb(blk_next) * L - OVL + n0 - in_shift;
        const bool next_fast = (blk_next < nblocks) && (s0n >= 0) && !(diag & 1);
        // diag 16: every wave fetches the same (cached) block -> separates HBM effects from the instruction stream's
        const int64_t s0f = (diag & 16) ? (int64_t)(lane & 0) : s0n;
        const srd_t nsrd = make_srd(in + (next_fast ? s0f : 0) * ISZ, next_fast ? (N - s0f) * ISZ : 0);
        // outputs beyond M are dropped by the descriptor's bounds check
        const int64_t obase = rb(blk) * LOUT;
        const srd_t osrd = make_srd(out + obase, (diag & 2) ? 0 : (M - obase) * 8);
        // filter-bank tails with an NCO: the block's share of the output rotation, phasor(phi0 + delta obase), wave-uniform;
        // parked in a per-wave LDS word until the tails need it (the 16-slot tail has neither SGPRs nor VGPRs to spare)
        if constexpr (NCO && (CHAN == 16 || CHAN == 9))
        {
            const float2 pb = phasor(nco_phi0 + nco_delta * (uint32_t)obase);
            if (lane == 0)
                *reinterpret_cast<cf *>(smem + LDS_QNCO + wid * 8) = (cf){pb.x, pb.y};
        }
        if constexpr (CHAN == 16)
        {
            // ---- 16-slot filter bank at the channel rate (decimation 16, round 3; tools/fft_model.py bank16) --------------
            // Channel s = the prototype moved up by s/16 cycles/sample; decimating by 16 aliases every slot centre to DC.
            // Folded spectrum of ALL 16 slots from one 16-point transform per group:
            //   Z_s(k0, k1) = sum_k2 H((k2 - s) mod 16) Y(k2) = FFT16(t * G0)[s],  G0[n2] = sum_k2 H(k2) W16^(n2 k2) (host table)
            // (round 4: the forward passes in (cos, tan) form; the b^n2 the inputs of this pass still carry is folded into the table)
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                cf t[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    t[j] = cmul_v<false>(r[phys(i, j)], hp[(i * 16 + j) * 64 + lane]);
                fft16<false>(t);
#pragma unroll
                for (int j = 0; j < 16; j++)
                    r[phys(i, j)] = t[j]; // slot (i, s = j)
            }
            // 256-point inverses, four slots at a time (cs = slot % 4 takes the place of mu2 in the 1024-point inverse):
            // lane = 4 mu1 + cs, slot mu0 -> y_s[16 mu0 + mu1], s = 4 b + cs; each lane stores to ITS channel's buffer
            constexpr int MU0_FIRST = OVL_ROWS / 4;
            constexpr int EARLY_B = I16 ? 3 : 1; // batches whose next-block rows are requested ahead of their inverse (no scratch)
            const int cs = lane & 3, mu1 = lane >> 2;
            // NCO (the context's NCO = a common fine offset of the whole slot grid):
            // output m = obase + 16 (mu0 - first) + mu1 is rotated by phasor(phi0 + delta m) = A(lane) * B(mu0 - first), B from
            // the table (step 16 delta, fft_build_tables)
            // with A(lane) = [phasor(phi0 + delta obase), wave-uniform, in SGPRs] * [phasor(delta mu1), table entries 32..47]
#pragma unroll
            for (int b = 0; b < 4; b++)
            {
                // a batch none of whose four slots is wanted is
                // not inverted; its registers are refilled with next-block rows all the same
                const bool wanted = ((chan.mask16 >> (4 * b)) & 15u) != 0u;
                cf a[16];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        a[4 * i + q] = r[phys(i, 4 * b + q)];
                // these 16 registers are dead: refill them with rows of the next block (the last batch after its inverse,
                // to keep the temporaries out of scratch)
                if (b < EARLY_B && next_fast)
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, 4 * b + q));
                }
                if (wanted)
                {
                    cf c[16];
                    inverse_tail256_tan(a, c, twe, xa, lane);
                    // this lane's channel: its buffer (pointer table in LDS) and the call-constant mix-down phase
                    // exp(-j 2 pi s (abs0 + n0) / 16) = W16^(s rot_e) (table entries 16..31: the 16th roots of unity)
                    float2 *po = reinterpret_cast<float2 *const *>(smem + LDS_QPTR)[4 * b + cs];
                    cf wl = ncob[16 + (((4 * b + cs) * (int)chan.rot_e) & 15)];
                    if constexpr (NCO)
                        wl = cmul_v<false>(cmul_v<false>(wl, *reinterpret_cast<const cf *>(smem + LDS_QNCO + wid * 8)), ncob[32 + mu1]);
                    const int64_t o0 = obase + mu1;
                    if (po != nullptr && !(diag & 2))
                        store_lane_rows<16 - MU0_FIRST, 16>(reinterpret_cast<cf *>(po) + o0, obase + LOUT <= M, o0, M, [&](int k) {
                            if constexpr (NCO)
                                return cmul_v<false>(c[MU0_FIRST + k], cmul_v<false>(wl, ncob[k]));
                            else
                                return cmul_v<false>(c[MU0_FIRST + k], wl);
                        });
                }
                if (b >= EARLY_B && next_fast)
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, 4 * b + q));
                }
            }
        }
        else if constexpr (CHAN == 2 || CHAN == 3)
        {
            // ---- decimate-by-2 tail (round 3): pass 3, multiply by H/4096, fold the 2 aliases
            // (k2 = k2' + 8 j) in place: r[phys(i, k2')] = z(i, k2'), k2' = 0..7; the other 8 registers of the group are dead
            // and refilled with rows of the next block right away
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
                    t[j] = cmul_v<false>(t[j], hp[(i * 16 + j) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < 8; j++)
                    r[phys(i, j)] = t[j] + t[j + 8];
                if (next_fast)
                {
#pragma unroll
                    for (int j = 8; j < 16; j++)
                        load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
            }
            // 2048-point inverse as TWO 1024-point inverses (even and odd outputs), so that every lane ends up with two
            // ADJACENT outputs and a store instruction writes 1 KiB contiguously (the first form of this tail, an 8-point
            // stage + the common tail twice, left each instruction with 32-byte pieces 64 bytes apart: 1.14 ms against 0.73 ms
            // of the selecting store, profiles/r03_decimate_2_tail.txt):
            //   y[2p]   = IFFT1024( Z[k] + Z[k + 1024] ),   y[2p+1] = IFFT1024( (Z[k] - Z[k + 1024]) conj W2048^k ),
            //   k = k0 + 16 k1 + 256 q, q = 0..3 (k + 1024 is k2' + 4);  W2048^k = W2048^(16 k1 + k0) W8^q
            // in place: r[phys(i, q)] = even spectrum, r[phys(i, q + 4)] = odd spectrum
            constexpr float R8 = 0.70710678118654752f;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const f2v wf = twf[i * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const cf u = r[phys(i, q)], v = r[phys(i, q + 4)];
                    r[phys(i, q)] = u + v;
                    cf d = cmul_v<true>(u - v, wf);
                    if (q == 1)
                        d = cmul_s<false>(d, (cf){R8, R8});
                    else if (q == 2)
                        d = (cf){-d.y, d.x}; // * (+j)
                    else if (q == 3)
                        d = cmul_s<false>(d, (cf){-R8, R8});
                    r[phys(i, q + 4)] = d;
                }
            }
            constexpr int MU0_FIRST = OVL_ROWS / 4; // 64 output PAIRS per mu0 slot = 256 input samples = 4 rows
            // NCO: outputs m = obase + 128 (mu0 - first) + 2 lane (+ 1) are rotated by phasor(phi0 + delta m): a lane factor
            // (formed here, ahead of the inverses, while registers are free), a wave-uniform step for the odd output, and the
            // row table (entries 64 full-rate rows = 32 outputs apart: entry 4 k)
            cf a_lane = {1.0f, 0.0f}, a_odd = {1.0f, 0.0f};
            if constexpr (NCO)
            {
                const float2 pa = phasor(nco_phi0 + nco_delta * ((uint32_t)obase + 2u * (uint32_t)lane));
                a_lane = (cf){pa.x, pa.y};
                const float2 ph = phasor(nco_delta);
                const cf odd = {__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ph.x))),
                                __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ph.y)))};
                a_odd = cmul_s<false>(a_lane, odd);
            }
            (void)a_lane; (void)a_odd;
            cf ce[16], co[16];
            {
                cf z[16];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        z[4 * i + q] = r[phys(i, q)];
                inverse_dec4(z, ce, twd, twe, xa, lane);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        z[4 * i + q] = r[phys(i, q + 4)];
                inverse_dec4(z, co, twd, twe, xa, lane);
            }
            if constexpr (NCO)
            {
#pragma unroll
                for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                {
                    ce[mu0] = cmul_v<false>(ce[mu0], cmul_v<false>(a_lane, ncob[4 * (mu0 - MU0_FIRST)]));
                    co[mu0] = cmul_v<false>(co[mu0], cmul_v<false>(a_odd, ncob[4 * (mu0 - MU0_FIRST)]));
                }
            }
            if constexpr (CHAN == 3)
            {
                // decimation 6, 10, ..., 62 (2 x odd): every sub-th output of this tail is a real output (KeepEvery, as behind the
                // decimate-by-4 tail); the pair of a lane never survives together, so two 8-byte stores with their own offsets.
                // The index arithmetic needs registers: the other half of the next block's rows is requested BEHIND the stores here.
                KeepEvery ke;
                ke.init(blk, (unsigned)LOUT, chan.sub);
                const int64_t qb = ke.qU + (ke.rem ? 1 : 0);
                const srd_t dsrd = make_srd(out + qb, (diag & 2) ? 0 : (decn_m - qb) * 8);
                const int lim = (int)((M - obase) < 65536 ? (M - obase) : 65536); // tail outputs of the call left from this block on
                int off = 2 * lane;
#pragma unroll
                for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                {
                    // (one running offset, made opaque: otherwise the 30 store offsets are computed ahead of the inverses and spill)
                    asm volatile("" : "+v"(off));
                    const int64_t k0 = ke.index((unsigned)off), k1 = ke.index((unsigned)off + 1u);
                    const unsigned so0 = (k0 >= 0 && off < lim) ? (unsigned)(k0 - qb) * 8u : 0xffffffffu;
                    const unsigned so1 = (k1 >= 0 && off + 1 < lim) ? (unsigned)(k1 - qb) * 8u : 0xffffffffu;
                    if constexpr (ACC) // second partition: add to what the first launch stored (a dropped lane reads 0)
                    {
                        ce[mu0] += buf_load(dsrd, so0, 0);
                        co[mu0] += buf_load(dsrd, so1, 0);
                    }
                    buf_store(dsrd, so0, 0, ce[mu0]);
                    buf_store(dsrd, so1, 0, co[mu0]);
                    off += 128;
                }
            }
            if (next_fast) // the other half of the next block's rows
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
            }
            unsigned vo128 = (unsigned)lane * 16u;
            if constexpr (CHAN == 2)
            {
#pragma unroll
            for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
            {
                typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));
                if constexpr (ACC) // second partition: add to what the first launch stored
                {
                    const u32x4s_t old = __builtin_amdgcn_raw_buffer_load_b128(osrd, vo128, 0, 0);
                    ce[mu0] += (cf){__uint_as_float(old[0]), __uint_as_float(old[1])};
                    co[mu0] += (cf){__uint_as_float(old[2]), __uint_as_float(old[3])};
                }
                const u32x4s_t w = {__float_as_uint(ce[mu0].x), __float_as_uint(ce[mu0].y), __float_as_uint(co[mu0].x),
                                    __float_as_uint(co[mu0].y)};
                // The row offset goes into the VECTOR offset, the scalar offset stays the literal 0: with a 16-byte store whose
                // row offset sat in an SGPR the compiler placed no wait state between the store and the next (inline-asm) VALU
                // write of its data registers, and the second dword of the data arrived corrupted now and then (found by the
                // chunked-equals-unchunked GPU test; tools/diag_dec2.py).  In this form it inserts the s_nop the hazard needs.
                // tools/check_store_hazard.py scans every unit's disassembly for the hazard at build time (csrc/Makefile);
                // IF_FIR_FFT_HAZARD_PROBE=1 compiles the old form (tests/test_host.py: the scanner must flag it).
#if defined(IF_FIR_FFT_HAZARD_PROBE) && IF_FIR_FFT_HAZARD_PROBE == 1
                __builtin_amdgcn_raw_buffer_store_b128(w, osrd, (unsigned)lane * 16u, (mu0 - MU0_FIRST) * 1024, IF_FIR_FFT_STORE_AUX);
#else
                __builtin_amdgcn_raw_buffer_store_b128(w, osrd, vo128, 0, IF_FIR_FFT_STORE_AUX);
#endif
                vo128 += 1024u;
                asm volatile("" : "+v"(vo128)); // one running offset register, not 15 precomputed ones
            }
            }
            (void)vo128;
        }
        else if constexpr (CHAN == 17)
        {
            // ---- filter bank at decimation 16, every channel at its own centre bin (round 4; the decimation-8 general form below at
            // the channel rate).  Centre bin B = 256 s + b: H_c(k) = H(k - B); with k = k_low + 256 k2 and k_low - b = kappa - 256 cy
            //   Z_c(k_low) = sum_n2 t[n2] W16^(n2 s') G0^kappa[n2],  s' = s + cy,  G0[n2] = sum_k2 H(k_low + 256 k2) W16^(n2 k2)
            // (t = the inputs of pass 3, G0 = the 16-slot bank's table, gathered from the lane that holds kappa); 256-point inverses,
            // four channels at a time: lane = 4 mu1 + ch, slot mu0 -> y_ch[16 mu0 + mu1].  The inputs still carry b_klow^n2 ((cos, tan)
            // forward passes); the gathered entry carries b_kappa^n2, and b_klow^n2 W16^(n2 s') = b_kappa^n2 W4096^(n2 B): the factor
            // beside the entry is wave-uniform (host, chan.tw, SGPRs), as in the decimation-8 general form.
            constexpr int MU0_FIRST = OVL_ROWS / 4;
            const int nch = (int)chan.count;
            const int lq = (lane >> 4) + 4 * (lane & 15); // k_low >> 2
            const f2v *rowt = reinterpret_cast<const f2v *>(smem + LDS_ROWT);
            for (int cq = 0; cq < nch; cq += 4)
            {
                const bool last = cq + 4 >= nch; // the inputs of pass 3 die with the last four channels: refill with the next block
                cf a[16];
                // (round 5: channel by channel, its 15 twiddles loaded once and in SGPRs before its first table gather; per group the
                // 16 gathers fly under the 15 products t[n2] W4096^(n2 B): gather_mac)
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    if (cq + ch >= nch) // fewer than four: empty quarters
                    {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            a[4 * i + ch] = (cf){0.f, 0.f};
                        continue;
                    }
                    const int cb = (int)chan.bin[cq + ch], b = cb & 255; // wave-uniform (the slot s = cb >> 8 is absorbed: see above)
                    cf tw[15];
#pragma unroll
                    for (int n2 = 1; n2 < 16; n2++)
                        tw[n2 - 1] = (cf){chan.tw[cq + ch][2 * (n2 - 1)], chan.tw[cq + ch][2 * (n2 - 1) + 1]}; // W4096^(n2 B)
                    // the lane that holds kappa: (k_low - b) >> 2 = lq - (b >> 2), one less in the groups below b % 4 (a borrow) -- two
                    // lanes per channel, picked per group by a wave-uniform test (round 4 recomputed them per group)
                    const int lk0 = (lq - (b >> 2)) & 63, lk1 = (lk0 - 1) & 63;
                    const int lane_k0 = ((lk0 & 3) << 4) | (lk0 >> 2), lane_k1 = ((lk1 & 3) << 4) | (lk1 >> 2);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const int ik = (i - b) & 3;                            // table group of kappa (kappa % 4)
                        const int lane_k = (i < (b & 3)) ? lane_k1 : lane_k0;
                        cf t[16];
#pragma unroll
                        for (int n2 = 0; n2 < 16; n2++)
                            t[n2] = r[phys(i, n2)];
                        a[4 * i + ch] = gather_mac<16, 64, IF_FIR_GM_NA, IF_FIR_GM_NB>(t, tw, hp + (ik * 16) * 64 + lane_k);
                    }
                }
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = 0; i < EARLY_GROUPS; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                cf c[16];
                inverse_tail256_tan(a, c, twe, xa, lane);
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = EARLY_GROUPS; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                // mix-down: output o = obase + 16 (mu0 - first) + mu1 of the call belongs to input sample abs0n0 + 16 o: a lane factor
                // (exact 32-bit phase) times the channel's row phasor (256 input samples per row)
                const int cs = lane & 3, mu1 = lane >> 2;
                const int cl = (cq + cs < nch) ? cq + cs : nch - 1;
                const int c1 = (cq + 1 < nch) ? cq + 1 : nch - 1, c2 = (cq + 2 < nch) ? cq + 2 : nch - 1, c3 = (cq + 3 < nch) ? cq + 3 : nch - 1;
                float2 *po = cs == 0 ? chan.out[cq] : cs == 1 ? chan.out[c1] : cs == 2 ? chan.out[c2] : chan.out[c3];
                const uint32_t pw = cs == 0 ? chan.pword[cq] : cs == 1 ? chan.pword[c1] : cs == 2 ? chan.pword[c2] : chan.pword[c3];
                const int64_t o0 = obase + mu1;
                const float2 pa = phasor(0u - pw * (chan.abs0n0 + 16u * (uint32_t)o0));
                const cf wl = {pa.x, pa.y};
                const f2v *rowp = rowt + cl * 16;
                // decimation 32, 48, 64 (16 x sub): every sub-th output of this tail is a real output (KeepEvery, as behind the
                // single-channel tails; sub = 1: all of them)
                if (chan.sub == 1u) // (wave-uniform: the plain decimation keeps its plain store loop -- the thinning costs it 5 %)
                {
                    if (cq + cs < nch && !(diag & 2))
                        store_lane_rows<16 - MU0_FIRST, 16>(reinterpret_cast<cf *>(po) + o0, obase + LOUT <= M, o0, M, [&](int k) {
                            return cmul_v<false>(c[MU0_FIRST + k], cmul_v<false>(wl, rowp[k]));
                        });
                }
                else if ((chan.sub & (chan.sub - 1u)) == 0u && chan.sub <= 16u)
                {
                    // decimation 32, 64 (sub = 2, 4 divides the 16 outputs of a row): a lane keeps all of its outputs or none, and the
                    // kept ones are 16 / sub apart -- the plain store loop with another step (round 5)
                    KeepEvery ke;
                    ke.init(blk, (unsigned)LOUT, chan.sub);
                    const int64_t kept0 = ke.index((unsigned)mu1); // of the lane's first output, or -1
                    if (cq + cs < nch && !(diag & 2) && kept0 >= 0)
                    {
                        cf *pl = reinterpret_cast<cf *>(po) + kept0;
                        const unsigned step = 16u / chan.sub;
                        const bool full = obase + LOUT <= M;
#pragma unroll
                        for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                        {
                            const cf v = cmul_v<false>(c[mu0], cmul_v<false>(wl, rowp[mu0 - MU0_FIRST]));
                            if (full || o0 + 16 * (mu0 - MU0_FIRST) < M)
                                __builtin_nontemporal_store(v, pl);
                            pl += step;
                        }
                    }
                }
                else
                {
                    KeepEvery ke;
                    ke.init(blk, (unsigned)LOUT, chan.sub);
                    if (cq + cs < nch && !(diag & 2))
                    {
#pragma unroll
                        for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                        {
                            const int64_t idx = o0 + 16 * (mu0 - MU0_FIRST);
                            const int64_t kept = ke.index((unsigned)mu1 + 16u * (unsigned)(mu0 - MU0_FIRST));
                            const cf v = cmul_v<false>(c[mu0], cmul_v<false>(wl, rowp[mu0 - MU0_FIRST]));
                            if (idx < M && kept >= 0)
                                __builtin_nontemporal_store(v, reinterpret_cast<cf *>(po) + kept);
                        }
                    }
                }
            }
        }
        else if constexpr (CHAN == 9)
        {
            // ---- filter bank at decimation 8, ALL SLOTS OF ONE PARITY (round 4; tools/fft_model.py bank8_parity) ----
            // Z_s(k2') = sum_a w_{k2'}[a] W16^(a s) G_q[a], q = (k2' - s) mod 2 (the per-channel form above).  For the slots of one parity,
            // s = 2 sigma + par, W16^(a s) = W16^(a par) W8^(a sigma): with the factor W16^(a par) in the table (a second image for
            // the odd slots, fft_build_tables bank_parity)
            //     Z_s(0) = FFT8( w0 . G_par )[sigma],      Z_s(1) = FFT8( w1 . G_(1 - par) )[sigma]
            // -- w0 and w1 are used ONCE each, so the 64 registers of the block turn into the 64 values Z_s(k2') (8 slots x 2 x 4 groups)
            // in place.  (All 16 slots at once would need 128 live values: the other parity is a second launch.)  16 multiplies
            // and two 8-point transforms per group serve eight channels: 88 packed instructions where the per-channel form spends
            // 8 x 60.  The inverses follow two slots at a time exactly as in the per-channel form; every lane stores to its slot's
            // buffer (pointer table in LDS, as in the 16-slot bank); slots nobody asked for are not inverted.
            // (the launcher passes the parity in `sub`; both parities in one launch: the virtual block's low bit -- then the ODD slots
            // take the even slots' image with the halves exchanged and the eight constants W16^a on top, so one image serves both)
            const int par = both ? (int)(blk & 1) : ((int)chan.sub & 1);
            const bool swap = both && par;
            const int h0 = swap ? 8 : 0, h1 = swap ? 0 : 8;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                cf t0[8], t1[8];
                // (cos, tan) form: the inputs of this pass still carry b^n2, b = W4096^(k0 + 16 k1) (passes 1 and 2 as in the
                // decimate-by-4 kernels).  t[a] b^a +- t[a + 8] b^(a + 8) = b^a (t[a] +- b^8 t[a + 8]): b^a sits in the table image,
                // b^8 = c (1 + j t) is the second of the three first-stage entries of pass 3 -- 3 packed FMAs per pair
                const cf e8 = tw1[(i * 3 + 1) * 64 + lane];
#pragma unroll
                for (int a8 = 0; a8 < 8; a8++)
                {
                    const cf u = r[phys(i, a8)], vb = tw_u<false>(r[phys(i, a8 + 8)], e8);
                    t0[a8] = cmul_v<false>(tw_ac<false>(u, vb, e8), hp[(i * 16 + h0 + a8) * 64 + lane]); // w0 . (b^a G_par W16^(a par)): first half of the image
                    t1[a8] = cmul_v<false>(tw_ac<true>(u, vb, e8), hp[(i * 16 + h1 + a8) * 64 + lane]);  // w1 . (b^a G_(1 - par) W16^(a par)): second half
                }
                if (swap) // (wave-uniform) W16^a, a = 1..7
                {
                    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
                    t0[1] = cmul_s<false>(t0[1], (cf){C1, -S1}); t1[1] = cmul_s<false>(t1[1], (cf){C1, -S1});
                    t0[2] = cmul_s<false>(t0[2], (cf){R, -R});   t1[2] = cmul_s<false>(t1[2], (cf){R, -R});
                    t0[3] = cmul_s<false>(t0[3], (cf){S1, -C1}); t1[3] = cmul_s<false>(t1[3], (cf){S1, -C1});
                    t0[4] = (cf){t0[4].y, -t0[4].x};             t1[4] = (cf){t1[4].y, -t1[4].x}; // -j
                    t0[5] = cmul_s<false>(t0[5], (cf){-S1, -C1}); t1[5] = cmul_s<false>(t1[5], (cf){-S1, -C1});
                    t0[6] = cmul_s<false>(t0[6], (cf){-R, -R});  t1[6] = cmul_s<false>(t1[6], (cf){-R, -R});
                    t0[7] = cmul_s<false>(t0[7], (cf){-C1, -S1}); t1[7] = cmul_s<false>(t1[7], (cf){-C1, -S1});
                }
                fft8<false>(t0);
                fft8<false>(t1);
#pragma unroll
                for (int sg = 0; sg < 8; sg++)
                {
                    r[phys(i, sg)] = t0[sg];     // Z_s(0), s = 2 sg + par
                    r[phys(i, sg + 8)] = t1[sg]; // Z_s(1)
                }
            }
            constexpr int MU0_FIRST = OVL_ROWS / 4;
            constexpr int EARLY_B = I16 ? 3 : 1; // batches whose next-block rows are requested ahead of their inverse (no scratch)
#pragma unroll
            for (int b = 0; b < 4; b++)
            {
                // slots of this batch: sigma = 2 b, 2 b + 1
                const unsigned want = (chan.mask16 >> (2 * (2 * b) + par)) & 1u, want1 = (chan.mask16 >> (2 * (2 * b + 1) + par)) & 1u;
                const bool wanted = (want | want1) != 0u;
                cf a[16];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int ch = 0; ch < 2; ch++)
                    {
                        const cf z0 = r[phys(i, 2 * b + ch)], z1 = r[phys(i, 8 + 2 * b + ch)];
                        a[4 * i + 2 * ch] = z0 + z1;
                        a[4 * i + 2 * ch + 1] = cmul_v<true>(z0 - z1, twd[(i * 4 + 2) * 64 + lane]); // conj W512^(16 k1 + k0)
                    }
                auto refill = [&]() {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int ch = 0; ch < 2; ch++)
                        {
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, 2 * b + ch));
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, 8 + 2 * b + ch));
                        }
                };
                if (b < EARLY_B && next_fast)
                    refill();
                if (wanted)
                {
                    cf c[16];
                    inverse_tail256_tan(a, c, twe, xa, lane);
                    // lane = 4 mu1 + 2 ch + mu2, slot mu0 -> y_s[32 mu0 + 2 mu1 + mu2], s = 2 (2 b + ch) + par.  Mix-down: the call
                    // constant W16^(s rot_e) (16th roots: table entries 16..31) times (-1)^(s m), m = obase + 32 (..) + 2 mu1 + mu2
                    // with obase even: the sign is (-1)^(s mu2)
                    const int sl = 2 * (2 * b + ((lane >> 1) & 1)) + par;
                    float2 *po = reinterpret_cast<float2 *const *>(smem + LDS_QPTR)[sl];
                    cf wl = ncob[16 + ((sl * (int)chan.rot_e) & 15)];
                    if ((sl & lane) & 1)
                        wl = (cf){-wl.x, -wl.y};
                    // NCO (the context's NCO = a common fine offset of the whole slot grid, as in the 16-slot tail): output m = obase +
                    // 32 (mu0 - first) + 2 mu1 + mu2 is also rotated by phasor(phi0 + delta m) = [block, per-wave LDS word] * [lane:
                    // table entries 32..63] * [row: entries 0..15]
                    if constexpr (NCO)
                        wl = cmul_v<false>(cmul_v<false>(wl, *reinterpret_cast<const cf *>(smem + LDS_QNCO + wid * 8)),
                                           ncob[32 + 2 * (lane >> 2) + (lane & 1)]);
                    const int64_t o0 = obase + 2 * (lane >> 2) + (lane & 1);
                    if (po != nullptr && !(diag & 2))
                        store_lane_rows<16 - MU0_FIRST, 32>(reinterpret_cast<cf *>(po) + o0, obase + LOUT <= M, o0, M, [&](int k) {
                            if constexpr (NCO)
                                return cmul_v<false>(c[MU0_FIRST + k], cmul_v<false>(wl, ncob[k]));
                            else
                                return cmul_v<false>(c[MU0_FIRST + k], wl);
                        });
                }
                if (b >= EARLY_B && next_fast)
                    refill();
            }
        }
        else if constexpr (CHAN == 8)
        {
            // ---- filter bank at decimation 8 (fs/16 slots, 2x oversampled channels; tools/fft_model.py bank8) ------------
            // First radix-2 stage of pass 3 once, in place: r[phys(i, a)] = w0[a] = t[a] + t[a+8], r[phys(i, a+8)] = w1[a].
            // Channel at slot s:  Z_s(k2') = sum_a w_{k2'}[a] W16^(a s) G_q[a],  q = (k2' - s) mod 2 (host table G, kernel
            // arguments W16^(a s)); 512-point inverse, two channels per small inverse (low = 2 ch + mu2):
            // lane = 4 mu1 + 2 ch + mu2, slot mu0 -> y_ch[32 mu0 + 2 mu1 + mu2]
            // (cos, tan) form (round 4): the inputs of this pass still carry b^n2, b = W4096^(k0 + 16 k1);
            // t[a] b^a +- t[a + 8] b^(a + 8) = b^a (t[a] +- b^8 t[a + 8]): b^8 = c (1 + j t) is applied here (three packed FMAs a pair), b^a is
            // folded into the table for the lane that OWNS the entry (see the general form below for lanes that read another lane's)
            static_assert(CHAN != 8 || TAN, "the decimation-8 bank is written for its (cos, tan) image");
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const cf e8 = tw1[(i * 3 + 1) * 64 + lane];
#pragma unroll
                for (int a8 = 0; a8 < 8; a8++)
                {
                    const cf u = r[phys(i, a8)], vb = tw_u<false>(r[phys(i, a8 + 8)], e8);
                    r[phys(i, a8)] = tw_ac<false>(u, vb, e8);
                    r[phys(i, a8 + 8)] = tw_ac<true>(u, vb, e8);
                }
            }
            constexpr int MU0_FIRST = OVL_ROWS / 4;
            const int nch = (int)chan.count;
            if constexpr (!NCO)
            {
            // ---- channels on the fs/16 slot grid, no NCO on the context (round 3's form: slot twiddles in SGPRs, the mix-down a sign)
            for (int cp = 0; cp < nch; cp += 2)
            {
                const bool last = cp + 2 >= nch; // the w values die with the last pair: refill with the next block
                cf a[16];
#pragma unroll
                for (int ch = 0; ch < 2; ch++)
                {
                    if (ch == 1 && cp + 1 >= nch) // odd count: an empty second half
                    {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                        {
                            a[4 * i + 2] = (cf){0.f, 0.f};
                            a[4 * i + 3] = (cf){0.f, 0.f};
                        }
                        continue;
                    }
                    const int c = cp + ch;
                    const int par = (int)chan.slot[c] & 1;
                    cf tw[7];
#pragma unroll
                    for (int a8 = 1; a8 < 8; a8++)
                        tw[a8 - 1] = (cf){chan.tw[c][2 * (a8 - 1)], chan.tw[c][2 * (a8 - 1) + 1]}; // W16^(a slot)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const f2v *g0 = hp + (i * 16 + 8 * par) * 64 + lane;       // k2' = 0: q = s & 1       (+ a * 64 entries)
                        const f2v *g1 = hp + (i * 16 + 8 * (1 - par)) * 64 + lane; // k2' = 1: q = (1 - s) & 1
                        cf t0[8], t1[8];
#pragma unroll
                        for (int a8 = 0; a8 < 8; a8++)
                        {
                            t0[a8] = r[phys(i, a8)];
                            t1[a8] = r[phys(i, a8 + 8)];
                        }
                        const cf z0 = gather_mac<8, 64, IF_FIR_GM_NA, IF_FIR_GM_NB>(t0, tw, g0); // (round 5: table reads ahead of the products)
                        const cf z1 = gather_mac<8, 64, IF_FIR_GM_NA, IF_FIR_GM_NB>(t1, tw, g1);
                        a[4 * i + 2 * ch] = z0 + z1;
                        a[4 * i + 2 * ch + 1] = cmul_v<true>(z0 - z1, twd[(i * 4 + 2) * 64 + lane]); // conj W512^(16 k1 + k0)
                    }
                }
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = 0; i < EARLY_GROUPS; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                cf c[16];
                inverse_tail256_tan(a, c, twe, xa, lane);
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = EARLY_GROUPS; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                // mix-down: exp(-j 2 pi s a / 16), a = abs0 + n0 + 8 m: the call constant rot0 times (-1)^(s m); m = obase +
                // 32 (mu0 - first) + 2 mu1 + mu2 with obase even, so the sign is (-1)^(s mu2)
                const int chl = (lane >> 1) & 1, cl = cp + chl;
                const int c1 = (cp + 1 < nch) ? cp + 1 : nch - 1;
                float2 *po = chl ? chan.out[c1] : chan.out[cp];
                const cf r0 = chl ? (cf){chan.rot0[c1][0], chan.rot0[c1][1]} : (cf){chan.rot0[cp][0], chan.rot0[cp][1]};
                const int sl = chl ? (int)chan.slot[c1] : (int)chan.slot[cp];
                cf wl = ((sl & lane) & 1) ? (cf){-r0.x, -r0.y} : r0;
                const int64_t o0 = obase + 2 * (lane >> 2) + (lane & 1);
                if (cl < nch && !(diag & 2))
                    store_lane_rows<16 - MU0_FIRST, 32>(reinterpret_cast<cf *>(po) + o0, obase + LOUT <= M, o0, M,
                                                        [&](int k) { return cmul_v<false>(c[MU0_FIRST + k], wl); });
            }
            }
            else
            {
            // ---- general form (template flag NCO): channels at any centre bin and / or an NCO on the context
            // Round 4: every channel has its own centre bin B = 256 s + b (the prototype moved up by B / 4096 cycles/sample: b = 0
            // is round 3's slot grid).  H_c(k) = H(k - B): with k = k_low + 256 k2 (k_low = k0 + 16 k1 < 256 held by this lane and
            // group) k_low - b = kappa - 256 cy, so the lane needs the table entries of low index kappa -- group (i - b) mod 4, the
            // lanes rotated: a permutation, conflict-free -- for the slot s' = s + cy (tools/fft_model.py bank8_bins):
            //   Z_c(k2') = sum_a w_{k2'}[a] W16^(a s') G_q^kappa[a],  q = (k2' - s') mod 2
            // The table entry of kappa carries b_kappa^a (folded in for its owner); this lane's data owe b_klow^a =
            // b_kappa^a W4096^(a b) W16^(-a cy), and W16^(a s') W16^(-a cy) = W16^(a s): the factor beside the gathered entry is the
            // WAVE-UNIFORM W4096^(a B), B = 256 s + b (host, chan.tw, in SGPRs as in the slot form) -- the lanes differ only in which
            // entry they read.
            // k_low >> 2 = lane / 16 + 4 (lane % 16) in every group
            const int lq = (lane >> 4) + 4 * (lane & 15);
            const f2v *rowt = reinterpret_cast<const f2v *>(smem + LDS_ROWT);
            for (int cp = 0; cp < nch; cp += 2)
            {
                const bool last = cp + 2 >= nch; // the w values die with the last pair: refill with the next block
                cf a[16];
                // (round 5: channel by channel, its seven twiddles in SGPRs before its first table gather; per group the 2 x 8 gathers
                // fly under the products w[a] W4096^(a B): gather_mac)
#pragma unroll
                for (int ch = 0; ch < 2; ch++)
                {
                    if (ch == 1 && cp + 1 >= nch) // odd count: an empty second half
                    {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                        {
                            a[4 * i + 2] = (cf){0.f, 0.f};
                            a[4 * i + 3] = (cf){0.f, 0.f};
                        }
                        continue;
                    }
                    const int c = cp + ch;
                    const int cb = (int)chan.bin[c], b = cb & 255, s = cb >> 8; // wave-uniform
                    cf tw[7];
#pragma unroll
                    for (int a8 = 1; a8 < 8; a8++)
                        tw[a8 - 1] = (cf){chan.tw[c][2 * (a8 - 1)], chan.tw[c][2 * (a8 - 1) + 1]}; // W4096^(a B)
                    // the lane that holds kappa and the parity of s' = s + cy: (k_low - b) >> 2 = lq - (b >> 2), one less in the groups below
                    // b % 4 (a borrow); negative: a borrow from k2 (cy = 1).  Two variants per channel, picked per group by a
                    // wave-uniform test (round 4 recomputed them per group): entry offsets of the halves q = s' & 1 and 1 - q
                    const int d0 = lq - (b >> 2), d1 = d0 - 1;
                    const int lk0 = d0 & 63, lk1 = d1 & 63;
                    const int par0 = (s + (d0 < 0 ? 1 : 0)) & 1, par1 = (s + (d1 < 0 ? 1 : 0)) & 1;
                    const int e0a = (((lk0 & 3) << 4) | (lk0 >> 2)) + 512 * par0, e0b = e0a + 512 - 1024 * par0;
                    const int e1a = (((lk1 & 3) << 4) | (lk1 >> 2)) + 512 * par1, e1b = e1a + 512 - 1024 * par1;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const int ik = (i - b) & 3;                                 // table group of kappa (kappa % 4)
                        const bool bor = i < (b & 3);
                        const f2v *g0 = hp + ik * 1024 + (bor ? e1a : e0a); // k2' = 0: q = s' & 1       (+ a * 64 entries)
                        const f2v *g1 = hp + ik * 1024 + (bor ? e1b : e0b); // k2' = 1: q = (1 - s') & 1
                        cf t0[8], t1[8];
#pragma unroll
                        for (int a8 = 0; a8 < 8; a8++)
                        {
                            t0[a8] = r[phys(i, a8)];
                            t1[a8] = r[phys(i, a8 + 8)];
                        }
                        const cf z0 = gather_mac<8, 64, IF_FIR_GM_NA, IF_FIR_GM_NB>(t0, tw, g0);
                        const cf z1 = gather_mac<8, 64, IF_FIR_GM_NA, IF_FIR_GM_NB>(t1, tw, g1);
                        a[4 * i + 2 * ch] = z0 + z1;
                        a[4 * i + 2 * ch + 1] = cmul_v<true>(z0 - z1, twd[(i * 4 + 2) * 64 + lane]); // conj W512^(16 k1 + k0)
                    }
                }
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = 0; i < EARLY_GROUPS; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                cf c[16];
                inverse_tail256_tan(a, c, twe, xa, lane);
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = EARLY_GROUPS; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                // mix-down: output o = obase + 32 (mu0 - first) + 2 mu1 + mu2 of the call belongs to input sample abs0n0 + 8 o and is
                // rotated by exp(-j 2 pi pword (abs0n0 + 8 o) / 2^32): a lane factor (exact 32-bit phase, one sincos per lane and
                // channel pair) times the channel's row phasor (LDS table, 256 input samples per row)
                const int chl = (lane >> 1) & 1, cl = cp + chl;
                const int c1 = (cp + 1 < nch) ? cp + 1 : nch - 1;
                float2 *po = chl ? chan.out[c1] : chan.out[cp];
                const uint32_t pw = chl ? chan.pword[c1] : chan.pword[cp];
                const int64_t o0 = obase + 2 * (lane >> 2) + (lane & 1);
                const float2 pa = phasor(0u - pw * (chan.abs0n0 + 8u * (uint32_t)o0));
                const cf wl = {pa.x, pa.y};
                const f2v *rowp = rowt + (chl ? c1 : cp) * 16;
                // decimation 24, 40, 56 (8 x sub): every sub-th output of this tail is a real output (KeepEvery; sub = 1: all of them)
                if (chan.sub == 1u) // (wave-uniform: decimation 8 itself keeps its plain store loop)
                {
                    if (cl < nch && !(diag & 2))
                        store_lane_rows<16 - MU0_FIRST, 32>(reinterpret_cast<cf *>(po) + o0, obase + LOUT <= M, o0, M, [&](int k) {
                            return cmul_v<false>(c[MU0_FIRST + k], cmul_v<false>(wl, rowp[k]));
                        });
                }
                else
                {
                    KeepEvery ke;
                    ke.init(blk, (unsigned)LOUT, chan.sub);
                    if (cl < nch && !(diag & 2))
                    {
#pragma unroll
                        for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                        {
                            const int64_t idx = o0 + 32 * (mu0 - MU0_FIRST);
                            const int64_t kept = ke.index((unsigned)(2 * (lane >> 2) + (lane & 1)) + 32u * (unsigned)(mu0 - MU0_FIRST));
                            if (idx < M && kept >= 0)
                            {
                                const cf v = cmul_v<false>(c[mu0], cmul_v<false>(wl, rowp[mu0 - MU0_FIRST]));
                                __builtin_nontemporal_store(v, reinterpret_cast<cf *>(po) + kept);
                            }
                        }
                    }
                }
            }
            }
        }
        else if constexpr (CHAN == 4)
        {
            // ---- uniform filter bank: one forward transform, one decimated inverse per channel ---------
            // Channel slot s = the prototype moved to s/16 cycles/sample and mixed down: H_s(k2) = H(k2 - s), so with the
            // merged table of the single-channel path  G_s[m0][q] = W16^(m0 s) * G[m0][(q - s) mod 4]  (DESIGN.md §3.7).
            // First radix-4 stage of pass 3 once, in place: r[phys(i, m0 + 4 q)] = y[q][m0].
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                cf t[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    t[j] = r[phys(i, j)];
                if constexpr (TAN)
                {
                    // inputs carry b^n2, b = W4096^(k0 + 16 k1); the b^m0 this stage still owes sits in the table G'
                    const cf e1 = tw1[(i * 3 + 0) * 64 + lane], e2 = tw1[(i * 3 + 1) * 64 + lane], e3 = tw1[(i * 3 + 2) * 64 + lane];
#pragma unroll
                    for (int m0 = 0; m0 < 4; m0++)
                        bfly4_tw<false>(t[m0], t[m0 + 4], t[m0 + 8], t[m0 + 12], e1, e2, e3, r[phys(i, m0)], r[phys(i, m0 + 4)],
                                        r[phys(i, m0 + 8)], r[phys(i, m0 + 12)]);
                }
                else
                {
#pragma unroll
                for (int m0 = 0; m0 < 4; m0++)
                    bfly4<false>(t[m0], t[m0 + 4], t[m0 + 8], t[m0 + 12], r[phys(i, m0)], r[phys(i, m0 + 4)],
                                 r[phys(i, m0 + 8)], r[phys(i, m0 + 12)]);
                }
            }
            constexpr int MU0_FIRST = OVL_ROWS / 4;
            const int nch = (int)chan.count;
            for (int ch = 0; ch < nch; ch++)
            {
                const bool last = (ch == nch - 1); // the y values die with the last channel: refill with the next block
                const int slot = (int)chan.slot[ch];
                const cf w1 = {chan.tw[ch][0], chan.tw[ch][1]}, w2 = {chan.tw[ch][2], chan.tw[ch][3]},
                         w3 = {chan.tw[ch][4], chan.tw[ch][5]};
                cf z[16];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    // (round 5: two sums at a time, their table reads requested ahead of the products: see the general form below)
#pragma unroll
                    for (int qp = 0; qp < 4; qp += 2)
                    {
                        cf gq[2][4], pr[2][4];
#pragma unroll
                        for (int q = 0; q < 2; q++)
                        {
                            const f2v *g = hp + (i * 16 + ((qp + q - slot) & 3)) * 64 + lane; // + m0 * 256 entries
#pragma unroll
                            for (int m0 = 0; m0 < 4; m0++)
                                gq[q][m0] = g[256 * m0];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 2; q++)
                        {
                            pr[q][0] = r[phys(i, 4 * (qp + q))];
                            pr[q][1] = cmul_s<false>(r[phys(i, 4 * (qp + q) + 1)], w1);
                            pr[q][2] = cmul_s<false>(r[phys(i, 4 * (qp + q) + 2)], w2);
                            pr[q][3] = cmul_s<false>(r[phys(i, 4 * (qp + q) + 3)], w3);
                        }
#pragma unroll
                        for (int q = 0; q < 2; q++)
                        {
                            const cf e = cmac_v(cmul_v<false>(pr[q][0], gq[q][0]), pr[q][2], gq[q][2]);
                            const cf o = cmac_v(cmul_v<false>(pr[q][1], gq[q][1]), pr[q][3], gq[q][3]);
                            z[4 * i + qp + q] = e + o;
                        }
                    }
                    if (last && i < EARLY_GROUPS && next_fast)
                    {
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                    }
                }
                cf c[16];
                if constexpr (TAN)
                    inverse_dec4_tan(z, c, twe, twd, xa, lane);
                else
                    inverse_dec4(z, c, twd, twe, xa, lane);
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = EARLY_GROUPS; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                // mix-down of the decimated output: exp(-j 2 pi slot a / 16), a = abs0 + n0 + 4 m, m = obase + 64 r + lane
                // with obase a multiple of 4: a call constant (rot0, host) times a quarter turn per lane
                const cf r0 = {chan.rot0[ch][0], chan.rot0[ch][1]};
                const int qt = (slot * lane) & 3;
                const cf wl = qt == 0 ? r0 : qt == 1 ? (cf){r0.y, -r0.x} : qt == 2 ? (cf){-r0.x, -r0.y} : (cf){-r0.y, r0.x};
                const srd_t csrd = make_srd(chan.out[ch] + obase, (diag & 2) ? 0 : (M - obase) * 8);
#pragma unroll
                for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                    buf_store(csrd, voff, (mu0 - MU0_FIRST) * 512, cmul_v<false>(c[mu0], wl));
            }
        }
        else if constexpr (CHAN == 5 || CHAN == 6)
        {
            // ---- filter bank at decimation 4, every channel at its own centre bin (round 4) ---------------------------------------
            // The slot form above with the table entries of kappa (k_low - b = kappa - 256 cy, gathered from the lane that holds kappa)
            // and the slot s' = s + cy:  Z_c(k_low, q) = sum_m0 y[q][m0] W16^(m0 s') G^kappa[m0][(q - s') mod 4].  The table holds
            // G' = b_kappa^m0 G (the factor the first stage owes, folded in for the lane that owns the entry); this lane's data owe
            // b_klow^m0 = b_kappa^m0 W4096^(m0 b) W16^(-m0 cy), and W16^(m0 s') W16^(-m0 cy) = W16^(m0 s): the factor beside the table is
            // the wave-uniform W4096^(m0 B), B = 256 s + b (host, chan.tw) -- the lanes differ only in WHICH entry they read.
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                cf t[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    t[j] = r[phys(i, j)];
                static_assert((CHAN != 5 && CHAN != 6) || TAN, "the general form at decimation 4 is written for the (cos, tan) image");
                const cf e1 = tw1[(i * 3 + 0) * 64 + lane], e2 = tw1[(i * 3 + 1) * 64 + lane], e3 = tw1[(i * 3 + 2) * 64 + lane];
#pragma unroll
                for (int m0 = 0; m0 < 4; m0++)
                    bfly4_tw<false>(t[m0], t[m0 + 4], t[m0 + 8], t[m0 + 12], e1, e2, e3, r[phys(i, m0)], r[phys(i, m0 + 4)],
                                    r[phys(i, m0 + 8)], r[phys(i, m0 + 12)]);
            }
            constexpr int MU0_FIRST = OVL_ROWS / 4;
            const int nch = (int)chan.count;
            const int lq = (lane >> 4) + 4 * (lane & 15); // k_low >> 2
            const f2v *rowt = reinterpret_cast<const f2v *>(smem + LDS_ROWT);
            for (int ch = 0; ch < nch; ch++)
            {
                const bool last = (ch == nch - 1); // the y values die with the last channel: refill with the next block
                const int cb = (int)chan.bin[ch], b = cb & 255, s = cb >> 8; // wave-uniform
                const cf w1 = {chan.tw[ch][0], chan.tw[ch][1]}, w2 = {chan.tw[ch][2], chan.tw[ch][3]},
                         w3 = {chan.tw[ch][4], chan.tw[ch][5]};
                cf z[16];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int ik = (i - b) & 3;                            // table group of kappa (kappa % 4)
                    const int d = lq - (b >> 2) - ((i < (b & 3)) ? 1 : 0); // (k_low - b) >> 2, negative: a borrow from k2
                    const int lk = d & 63;
                    const int lane_k = ((lk & 3) << 4) | (lk >> 2);        // the lane that holds kappa in group ik
                    const int sp = s + (d < 0 ? 1 : 0);
                    // (round 5: two of the group's four sums at a time -- their eight gathers are requested first, the six products
                    // y w fly under them, then the multiply-accumulates; round 4's form had every table read in front of its use)
#pragma unroll
                    for (int qp = 0; qp < 4; qp += 2)
                    {
                        cf gq[2][4], pr[2][4];
#pragma unroll
                        for (int q = 0; q < 2; q++)
                        {
                            const f2v *g = hp + (ik * 16 + ((qp + q - sp) & 3)) * 64 + lane_k; // + m0 * 256 entries
#pragma unroll
                            for (int m0 = 0; m0 < 4; m0++)
                                gq[q][m0] = g[256 * m0];
                        }
                        __builtin_amdgcn_sched_barrier(0); // (the gathers stay up here: see gather_mac)
#pragma unroll
                        for (int q = 0; q < 2; q++)
                        {
                            pr[q][0] = r[phys(i, 4 * (qp + q))];
                            pr[q][1] = cmul_s<false>(r[phys(i, 4 * (qp + q) + 1)], w1);
                            pr[q][2] = cmul_s<false>(r[phys(i, 4 * (qp + q) + 2)], w2);
                            pr[q][3] = cmul_s<false>(r[phys(i, 4 * (qp + q) + 3)], w3);
                        }
#pragma unroll
                        for (int q = 0; q < 2; q++)
                        {
                            const cf e = cmac_v(cmul_v<false>(pr[q][0], gq[q][0]), pr[q][2], gq[q][2]);
                            const cf o = cmac_v(cmul_v<false>(pr[q][1], gq[q][1]), pr[q][3], gq[q][3]);
                            z[4 * i + qp + q] = e + o;
                        }
                    }
                    if (last && i < EARLY_GROUPS && next_fast)
                    {
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                    }
                }
                cf c[16];
                inverse_dec4_tan(z, c, twe, twd, xa, lane);
                if (last && next_fast)
                {
#pragma unroll
                    for (int i = EARLY_GROUPS; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                // mix-down: output o = obase + 64 (mu0 - first) + lane of the call belongs to input sample abs0n0 + 4 o: a lane factor
                // (exact 32-bit phase) times the channel's row phasor (256 input samples per row)
                const float2 pa = phasor(0u - chan.pword[ch] * (chan.abs0n0 + 4u * (uint32_t)(obase + lane)));
                const cf wl = {pa.x, pa.y};
                const f2v *rowp = rowt + ch * 16;
                // decimation 12, 20, 28, ... (4 x sub): every sub-th output of this tail is a real output (KeepEvery, as in the
                // single-channel tail; sub = 1: all of them).  Descriptor over the kept outputs from this block's first one on.
                if constexpr (CHAN == 5) // (decimation 4 itself; its own instantiation: with both store loops in one kernel it ran 4 % slower)
                {
                    const srd_t csrd = make_srd(chan.out[ch] + obase, (diag & 2) ? 0 : (M - obase) * 8);
#pragma unroll
                    for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                        buf_store(csrd, voff, (mu0 - MU0_FIRST) * 512, cmul_v<false>(c[mu0], cmul_v<false>(wl, rowp[mu0 - MU0_FIRST])));
                }
                else
                {
                    KeepEvery ke;
                    ke.init(blk, (unsigned)LOUT, chan.sub);
                    const int64_t qb = ke.qU + (ke.rem ? 1 : 0);
                    const srd_t csrd = make_srd(chan.out[ch] + qb, (diag & 2) ? 0 : (decn_m - qb) * 8);
                    const int lim = (int)((M - obase) < 65536 ? (M - obase) : 65536); // tail outputs of the call left from this block on
#pragma unroll
                    for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                    {
                        const int64_t kept = ke.index((unsigned)lane + 64u * (unsigned)(mu0 - MU0_FIRST));
                        const unsigned so = (kept >= 0 && lane + 64 * (mu0 - MU0_FIRST) < lim) ? (unsigned)(kept - qb) * 8u : 0xffffffffu;
                        buf_store(csrd, so, 0, cmul_v<false>(c[mu0], cmul_v<false>(wl, rowp[mu0 - MU0_FIRST])));
                    }
                }
            }
        }
        else if constexpr (DEC4)
        {
            // ---- pass 3, multiply by H/4096, fold the 4 aliases: z(i, k2') = sum_j Y(i, k2' + 4j) ------------------
            cf z[16];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                cf t[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    t[j] = r[phys(i, j)];
                // Only the first radix-4 stage of the 16-point transform is computed: y[q][m0] = sum_m1 t[m0 + 4 m1] W4^(m1 q).
                // Its second stage, the multiplication by H and the alias fold are one linear map per output,
                //   z(i, q) = sum_p H(q + 4p) Y(q + 4p) = sum_m0 y[q][m0] * G[m0][q],
                //   G[m0][q] = W16^(m0 q) * sum_p H(q + 4p) W4^(m0 p)      (host table, fft_build_tables)
                // i.e. 16 complex MACs instead of 4 butterflies + 8 twiddles + 16 multiplies + 12 adds.
                cf y[4][4];
                if constexpr (TAN)
                {
                    // (cos, tan) form: the inputs carry b^n2, b = W4096^(k0 + 16 k1) (pass 1's and pass 2's twiddles, moved here);
                    // this stage applies b^4, b^8, b^12 inside its butterflies, the b^m0 it still owes sits in the table (G')
                    const cf e1 = tw1[(i * 3 + 0) * 64 + lane], e2 = tw1[(i * 3 + 1) * 64 + lane], e3 = tw1[(i * 3 + 2) * 64 + lane];
#pragma unroll
                    for (int m0 = 0; m0 < 4; m0++)
                        bfly4_tw<false>(t[m0], t[m0 + 4], t[m0 + 8], t[m0 + 12], e1, e2, e3, y[0][m0], y[1][m0], y[2][m0], y[3][m0]);
                }
                else
                {
#pragma unroll
                for (int m0 = 0; m0 < 4; m0++)
                    bfly4<false>(t[m0], t[m0 + 4], t[m0 + 8], t[m0 + 12], y[0][m0], y[1][m0], y[2][m0], y[3][m0]);
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    cf acc = cmul_v<false>(y[q][0], hp[(i * 16 + q) * 64 + lane]);
#pragma unroll
                    for (int m0 = 1; m0 < 4; m0++)
                        acc = cmac_v(acc, y[q][m0], hp[(i * 16 + m0 * 4 + q) * 64 + lane]);
                    z[4 * i + q] = acc;
                }
                // the 16 registers of this group are dead: refill them with rows of the next block right away, so the
                // loads have the rest of pass 3 and the whole small inverse to land (EARLY_GROUPS of the 4 batches;
                // the last ones are issued after the inverse to keep its temporaries out of scratch)
                if (i < EARLY_GROUPS && next_fast)
                {
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
                }
                }
            FFT_STAMP(5);
            cf c[16];
            if constexpr (TAN)
                inverse_dec4_tan(z, c, twe, twd, xa, lane);
            else
                inverse_dec4(z, c, twd, twe, xa, lane);
            FFT_STAMP(6);
            if (next_fast)
            {
#pragma unroll
                for (int i = EARLY_GROUPS; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        load_row<I16, LAUX, EDGE>(r, nsrd, lane, phys(i, j));
            }
            constexpr int MU0_FIRST = OVL_ROWS / 4; // first valid 64-output row of the decimated block
            if constexpr (NCO)
            {
                // SPEC §3.2: output m = obase + 64 r + lane is rotated by phasor(phi0 + delta m) = A(lane) * B(r)
                const float2 pa = phasor(nco_phi0 + nco_delta * ((uint32_t)obase + (uint32_t)lane));
                const cf a_lane = {pa.x, pa.y};
#pragma unroll
                for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                    c[mu0] = cmul_v<false>(c[mu0], cmul_v<false>(a_lane, ncob[mu0 - MU0_FIRST]));
            }
            if constexpr (CHAN == 1)
            {
                // decimation 8, 12, ..., 64: every sub-th output of the decimate-by-4 tail is a real output.  Descriptor over the
                // kept outputs from this block's first one on (decn_m of them in the call); a lane that keeps nothing, or an
                // index beyond the end, is dropped by the bounds check.
                KeepEvery ke;
                ke.init(blk, (unsigned)LOUT, chan.sub);
                const int64_t qb = ke.qU + (ke.rem ? 1 : 0); // ceil(obase / sub): the first kept output of this block, wave-uniform
                const srd_t dsrd = make_srd(out + qb, (diag & 2) ? 0 : (decn_m - qb) * 8);
                const int lim = (int)((M - obase) < 65536 ? (M - obase) : 65536); // tail outputs of the call left from this block on
#pragma unroll
                for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                {
                    const int64_t kept = ke.index((unsigned)lane + 64u * (unsigned)(mu0 - MU0_FIRST));
                    const unsigned so = (kept >= 0 && lane + 64 * (mu0 - MU0_FIRST) < lim) ? (unsigned)(kept - qb) * 8u : 0xffffffffu;
                    if constexpr (ACC) // second partition: add to what the first launch stored (a dropped lane reads 0)
                        c[mu0] += buf_load(dsrd, so, 0);
                    buf_store(dsrd, so, 0, c[mu0]);
                }
            }
            else
            {
#pragma unroll
                for (int mu0 = MU0_FIRST; mu0 < 16; mu0++)
                {
                    if constexpr (ACC) // second partition: add to what the first launch stored
                        c[mu0] += buf_load(osrd, voff, (mu0 - MU0_FIRST) * 512);
                    buf_store(osrd, voff, (mu0 - MU0_FIRST) * 512, c[mu0]);
                }
            }
        }
        else
        {
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                cf t[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    t[j] = r[phys(i, j)];
                if constexpr (TANF) // inputs carry b^n2, b = W4096^(k0 + 16 k1): second-stage entries at T[k0 + 16 k1 + 256 q]
                    fft16_tw_T<false, 64>(t, tw1 + i * 3 * 64 + lane, tt + (t_fwd3 ^ (unsigned)i));
                else
                    fft16<false>(t);
                // ---- pointwise multiply by H/4096 and start the inverse (pass 3^-1) in the same registers -------
#pragma unroll
                for (int j = 0; j < 16; j++)
                    t[j] = cmul_v<false>(t[j], hp[(i * 16 + j) * 64 + lane]);
                fft16<true>(t);
#pragma unroll
                for (int j = 0; j < 16; j++)
                    r[phys(i, j)] = t[j];
                }
            exchange2(r, xa);
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                cf t[16];
                if constexpr (TANF)
                {
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        t[j] = r[phys(i, j)];
                    fft16_tw<true, 16>(t, twd + (lane & 15)); // inputs carry conj(W256^n2)^k1, n2 = lane % 16
                }
                else
                {
#pragma unroll
                for (int j = 0; j < 16; j++)
                {
                    if (j == 0)
                        t[j] = r[phys(i, j)];
                    else
                        t[j] = cmul_v<true>(r[phys(i, j)], tw2[j * 16 + (lane & 15)]);
                }
                fft16<true>(t);
                }
#pragma unroll
                for (int j = 0; j < 16; j++)
                    r[phys(i, j)] = t[j];
                }
            exchange1_inv(r);
            cf a_lane = {1.0f, 0.0f};
            if constexpr (NCO)
            {
                const float2 pa = phasor(nco_phi0 + nco_delta * ((uint32_t)obase + (uint32_t)lane));
                a_lane = (cf){pa.x, pa.y};
            }
            (void)a_lane;
            // general decimation (DECN): the block is filtered at full rate; full-rate output n = obase + 64 r + lane is
            // kept if n - decn_n0 is a non-negative multiple of D, as output (n - decn_n0) / D.  One 64-bit division per
            // block and lane (row 0); a row then adds 64 r < 4096 to the remainder, divided by D <= 64 with an exact
            // multiply-shift (ceil(2^18 / D), exact for numerators below 2^12).
            unsigned drho0 = 0, dmagic = 0;
            int drel0 = 0;
            srd_t dsrd = osrd;
            if constexpr (DECN)
            {
                const uint64_t tp = (uint64_t)(obase + lane - decn_n0 + decn); // > 0 because decn_n0 < D
                const uint64_t qq = tp / (uint32_t)decn;
                drho0 = (unsigned)(tp - qq * (uint32_t)decn);
                const int64_t q0 = (int64_t)qq - 1; // floor((n - n0) / D): -1 for the samples ahead of the first output
                const uint64_t tpb = (uint64_t)(obase - decn_n0 + decn);
                const int64_t qfirst = (int64_t)(tpb / (uint32_t)decn) - 1; // lane 0's, wave-uniform by construction
                const int64_t qb = qfirst < 0 ? 0 : qfirst;
                drel0 = (int)(q0 - qb);
                dmagic = (262144u + (unsigned)decn - 1u) / (unsigned)decn;
                dsrd = make_srd(out + qb, (diag & 2) ? 0 : (decn_m - qb) * 8);
            }
            (void)drho0; (void)dmagic; (void)drel0; (void)dsrd;
            // ---- last inverse pass, group by group: finish 16 rows, store them, and refill the same registers with
            //      the next block's rows (the loads fly while the remaining groups and the next forward pass compute)
#pragma unroll
            for (int rho = 0; rho < 4; rho++)
            {
                cf t[16];
                if constexpr (TANF)
                {
                    // inputs carry conj(b)^k0, b = W4096^(lane + 64 rho): first-stage entries (b^4, b^8, b^12) at T[4 lane + 256 rho],
                    // second-stage entries at T[lane + 64 rho + 256 q]
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        t[j] = r[4 * j + rho];
                    fft16_tw_T<true, 1024>(t, tt + t_inv1s + 256 * rho, tt + ((t_inv1 ^ (unsigned)(((rho & 1) << 1) | ((rho >> 1) << 3))) + 64 * rho));
                }
                else
                {
#pragma unroll
                for (int j = 0; j < 16; j++)
                {
                    if (j == 0)
                        t[j] = r[4 * j + rho];
                    else
                        t[j] = cmul_v<true>(r[4 * j + rho], tw1[(rho * 16 + j) * 64 + lane]);
                }
                fft16<true>(t);
                }
#pragma unroll
                for (int j = 0; j < 16; j++)
                {
                    const int row = 4 * j + rho;
                    if (row >= OVL_ROWS)
                    {
                        if constexpr (NCO)
                            t[j] = cmul_v<false>(t[j], cmul_v<false>(a_lane, ncob[row - OVL_ROWS]));
                        if constexpr (DECN)
                        {
                            const unsigned u = drho0 + 64u * (unsigned)(row - OVL_ROWS);
                            const unsigned qd = (u * dmagic) >> 18;
                            const int rel = drel0 + (int)qd;
                            const bool keep = (u - qd * (unsigned)decn == 0u) && rel >= 0;
                            const unsigned so = keep ? (unsigned)rel * 8u : 0xffffffffu; // out of range = dropped
                            if constexpr (ACC)
                                t[j] += buf_load(dsrd, so, 0);
                            buf_store(dsrd, so, 0, t[j]);
                        }
                        else
                        {
                            if constexpr (ACC)
                                t[j] += buf_load(osrd, voff, (row - OVL_ROWS) * 512);
                            buf_store(osrd, voff, (row - OVL_ROWS) * 512, t[j]);
                        }
                    }
                }
                if (next_fast)
                {
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        load_row<I16, LAUX, EDGE>(r, nsrd, lane, 4 * j + rho);
                }
                }
        }
        FFT_STAMP(7);
        dbg_it++;
        loaded = next_fast;
        blk = blk_next;
    }
#ifndef IF_FIR_FFT_STAMPS
    if (dbg && lane == 0)
    {
        unsigned long long *d = dbg + 4 * ((size_t)blockIdx.x * FFT_WAVES + wid);
        d[0] = st_r0;
        d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = st_c0;
        d[3] = __builtin_amdgcn_s_memtime();
    }
#endif
}

#endif // IF_FIR_FFT_ROWS
#ifdef IF_FIR_FFT_HOST // ================= host side =================
// Host view of the block queue (see queue_take): groups of FFT_WAVES blocks in global order; workgroup b starts with
// global group b (static), every further group of a workgroup is global group wgs + ticket.  Tickets keep being drawn
// past the end (a wave learns that it is done by receiving a block >= nblocks), at most one per group slot 0 taken, so a
// launch draws fewer than groups + 2 * wgs of them; the counter is re-zeroed by the launch before.
void fft_schedule(int64_t nblocks, int64_t wgs_max, FftSchedule &s)
{
    const int64_t groups = (nblocks + FFT_WAVES - 1) / FFT_WAVES;
    s.RA = FFT_WAVES; // blocks per group
    s.nA = groups;
    s.RB = Q_AHEAD;   // static groups per workgroup = groups fetched ahead
    s.nB = 0;
    s.wgs = groups < wgs_max ? groups : wgs_max;
    if (s.wgs < 1)
        s.wgs = 1;
    s.tickets = groups + 2 * s.wgs; // upper bound of the counter at the end of the launch
}

#endif
#ifdef IF_FIR_FFT_ROWS
template <int OVL_ROWS, bool DEC4, bool I16, bool NCO, int CHAN = 0, bool DECN = false, bool ACC = false>
static hipError_t launch_fft_t(const LaunchArgs &a)
{
    auto kern = fir_fft_kernel<OVL_ROWS, DEC4, I16, NCO, CHAN, DECN, ACC>;
    constexpr int L = FFT_N - 64 * OVL_ROWS;
    constexpr int LOUT = (CHAN == 16 || CHAN == 17) ? L / 16 : (CHAN == 8 || CHAN == 9) ? L / 8 : (CHAN == 2 || CHAN == 3) ? L / 2 : DEC4 ? L / 4 : L;
    static DeviceSetup setup;
    int ncus = 0;
    {
        const hipError_t e = device_setup(setup, a.device, reinterpret_cast<const void *>(kern), FFT_LDS_BYTES, &ncus);
        if (e != hipSuccess)
            return e;
    }
    // DECN: the kernel runs at full rate over the N inputs (blocks, run queue, history as for D = 1) and keeps every
    // D-th output, the first one at full-rate index n0
    // (decimation 4 sub behind the decimate-by-4 tail, CHAN == 1: the tail runs at the fs/4 rate and keeps every sub-th output)
    constexpr int F = (CHAN == 16 || CHAN == 17) ? 16 : (CHAN == 8 || CHAN == 9) ? 8 : (CHAN == 2 || CHAN == 3) ? 2 : DEC4 ? 4 : 1; // the tail's own decimation
    ChanArgs ca = a.chan ? *a.chan : ChanArgs{};
    ca.sub = (CHAN == 1 || CHAN == 5 || CHAN == 6) ? (uint32_t)(a.D / 4) : CHAN == 3 ? (uint32_t)(a.D / 2) : CHAN == 9 ? (ca.sub & 3u) /* bit 0: the parity, bit 1: both parities */
             : (CHAN == 8 && NCO) ? (uint32_t)(a.D / 8) : CHAN == 17 ? (uint32_t)(a.D / 16) : 1u; // (general bank forms: D = F x sub)
    const int64_t m_rate = DECN ? a.N : CHAN == 9 ? a.M : (a.M - 1) * (int64_t)ca.sub + 1; // (CHAN 9: `sub` carries the slot parity)
    const int32_t n0_rate = DECN ? 0 : a.n0;
    const int64_t nblocks = (a.M > 0 ? (m_rate + LOUT - 1) / LOUT : 0) * ((CHAN == 9 && (ca.sub & 2u)) ? 2 : 1); // (both parities: virtual blocks)
    if (nblocks <= 0)
        return hipSuccess;
    const int64_t wgs_max = (a.grid_limit > 0 && a.grid_limit < ncus) ? a.grid_limit : ncus;
    FftSchedule sch;
    fft_schedule(nblocks, wgs_max, sch);
    const int64_t wgs = sch.wgs;
    // groups and tail (if_fir_fft_queue.h): a remainder of at most one block per SIMD is kept out of the groups; diag 256
    // (development) switches the tail off
    // (diag 2048, development: the tail phase in launches of up to 16 two-wave rounds)
    const int64_t nblocks_main = (a.diag & 256) ? nblocks : queue_main_blocks(nblocks, wgs, (a.diag & 2048) ? 16 : Q_TAIL_MAX_ROUNDS);
    // two global counters used alternately: a launch draws from one and zeroes the other for the launch behind it
    // (same stream, so it has finished before that one starts); after anybody else touched the words, start over
    uint32_t qsel = 0;
    if (a.queue_base && a.queue_valid && *a.queue_valid)
        qsel = *a.queue_base & 1u;
    else
    {
        hipError_t e = hipMemsetAsync(a.queue, 0, 16, a.stream);
        if (e != hipSuccess)
            return e;
    }
    if (a.queue_base && a.queue_valid)
    {
        *a.queue_base = qsel ^ 1u;
        *a.queue_valid = true;
    }
    chan_arg_t<CHAN> cak;
    if constexpr (CHAN >= 4)
        cak = ca;
    else
        cak.sub = ca.sub;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(512), FFT_LDS_BYTES, a.stream,
                       reinterpret_cast<const f2v *>(a.in), reinterpret_cast<f2v *>(a.out),
                       reinterpret_cast<const f2v *>(a.fft_tables), reinterpret_cast<const f2v *>(a.hist_full), a.hist_len, a.N,
                       n0_rate, m_rate, nblocks, nblocks_main, (unsigned int *)a.queue,
                       (unsigned long long *)a.dbg, (int32_t)a.diag,
                       DECN ? 0u - a.nco_word * a.nco_abs0 : nco_phi0(a),
                       DECN ? 0u - a.nco_word : 0u - a.nco_word * (uint32_t)F,
                       cak, qsel, a.hist_out, (int32_t)a.D, (int32_t)a.n0, a.M,
                       (int32_t)a.in_shift);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess && a.queue_valid)
        *a.queue_valid = false; // nothing ran: the counters are in an unknown state
    return le;
}

#endif // IF_FIR_FFT_ROWS
#ifdef IF_FIR_FFT_HOST
// D = 1 and D = 4 have their own kernels; any other decimation runs the full-rate kernel with a selecting store.
// Taps: the first T-1 outputs of a 4096-point block are discarded, in whole 64-sample rows (4, 8, 16, 32 or 48 of the
// 64): up to 257 taps cost 6 % of the block, 513 taps 12.5 %, 1025 taps 25 %, 2049 taps half, 3073 taps three quarters.
// Routing of a decimation-8 filter-bank call whose channels sit on the slot grid (launch_fft_rows; one definition for the launcher and
// the CPU test): a slot parity with at least four channels, none of the call's slots listed twice, is served by ONE all-slots launch
// (pmask[parity] = its slots, else 0); `rest` = bit c set for every channel c left to the per-channel form.
void fft_bank8_plan(const uint32_t *slots, uint32_t count, bool all_slots_available, uint32_t pmask[2], uint32_t *rest)
{
    uint32_t seen = 0, m[2] = {0, 0};
    int npar[2] = {0, 0};
    bool dup = false;
    for (uint32_t c = 0; c < count; c++)
    {
        const uint32_t sl = slots[c] & 15u;
        dup = dup || ((seen >> sl) & 1u);
        seen |= 1u << sl;
        m[sl & 1u] |= 1u << sl;
        npar[sl & 1u]++;
    }
    *rest = 0;
    for (int par = 0; par < 2; par++)
        pmask[par] = (all_slots_available && !dup && npar[par] >= 4) ? m[par] : 0u;
    for (uint32_t c = 0; c < count; c++)
        if (!((pmask[slots[c] & 1u] >> (slots[c] & 15u)) & 1u))
            *rest |= 1u << c;
}

// The filter bank's tail for a decimation: 4, 8, 16 themselves; channels at their own centres (`general`) also every other multiple
// of 4 up to 64 -- the largest of 16, 8, 4 that divides it, the tail then keeps every (D / F)-th output.  0: not served.
int fft_bank_tail(int D, bool general)
{
    if (D == 4 || D == 8 || D == 16)
        return D;
    if (!general || D < 4 || D > 64 || (D & 3))
        return 0;
    return (D % 16 == 0) ? 16 : (D % 8 == 0) ? 8 : 4;
}

bool fft_supported(int T, int D)
{
    return D >= 1 && D <= 64 && T >= 1 && T <= 4096;
}

// Which decimations have a decimating tail (frequency-domain alias fold + small inverse): every EVEN one, D = F * sub with F the
// tail's own decimation.  F = 4: decimation 4 and every other multiple of 4 up to 64 -- the decimate-by-4 tail keeping every
// sub-th output (round 3; measured faster than the one-channel filter-bank tails at 8 / 16 it replaced for single channels,
// profiles/r03_composite_decimations.txt).  F = 2: decimation 2, and 6, 10, ..., 62 the same way behind the decimate-by-2 tail.
// Odd decimations run the full-rate kernel with a selecting store.  Filters of 3074..4096 taps (two partitions) take the same
// tails, the second partition accumulating.  One definition for the launcher, the shim's tables and the multi-channel
// front's chunk grid.
bool fft_tail(int T, int D, int *pF, int *pSub)
{
    int F = 1;
    if (D >= 1 && D <= 64 && T >= 1 && T <= 4096)
        F = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1;
    if (pF)
        *pF = F;
    if (pSub)
        *pSub = D / F;
    return F > 1;
}

// 3074..4096 taps: two partitions of at most FFT_PART taps each, y = h_a * x + h_b * (x delayed by FFT_PART)
bool fft_two_partitions(int T)
{
    return T > 3073;
}

// Odd decimations divisible by 3 or 5 (round 4, fir_odd_kernel): D = F sub; a block of F x 1024 input samples gives 1024 outputs
// at the fs/F rate, the first ceil((T - 1 + F - 1) / F) of which are invalid -- dropped as 2, 4 or 8 rows of 64.
bool fft_odd_tail(int T, int D, int *pF, int *pSub, int *pOvlr)
{
    int F = 1, ovlr = 0;
    if (D >= 3 && D <= 64 && (D & 1) && T >= 1 && !fft_two_partitions(T))
    {
        // (F = 5 -- decimation 5, 25, 35, 55 -- was written and dropped: five phase streams of 16 registers + the transforms'
        // temporaries do not fit 256 VGPRs, the compiler spilled 112 of them; those decimations keep the selecting store)
        F = (D % 3 == 0) ? 3 : 1;
        if (F > 1)
        {
            const int need = (T - 1 + F - 1 + F - 1) / F; // outputs of a block that see samples ahead of it
            // (8 dropped rows -- up to 1535 taps -- were built and measured 18 % slower than the selecting store: half of every
            // block is overlap, profiles/r04_odd_decimation.txt)
            ovlr = need <= 128 ? 2 : need <= 256 ? 4 : 0;
            if (!ovlr)
                F = 1;
        }
    }
    if (pF)
        *pF = F;
    if (pSub)
        *pSub = F > 1 ? D / F : 1;
    if (pOvlr)
        *pOvlr = ovlr;
    return F > 1;
}

// Overlap rows of the (taps, decimation) pair.  (Round 4 built and measured a 2-row kernel, L = 3968, for filters of at most 129
// taps on the full-rate pipeline -- 16 913 instead of 17 477 blocks for BASELINE configs[1]: within 1 % of the 4-row kernel on a
// stream that is not re-read from the memory-side cache, profiles/r04_two_row_overlap.txt -- and removed it again.)
int fft_overlap_rows(int T, int D)
{
    if (fft_two_partitions(T))
        return 32; // each partition runs the 32-row kernel
    (void)D;
    return (T - 1 <= 256) ? 4 : (T - 1 <= 512) ? 8 : (T - 1 <= 1024) ? 16 : (T - 1 <= 2048) ? 32 : 48;
}

// new input samples per block of the overlap-save kernel for this filter: streams cut at multiples of it (and of the
// decimation) give bit-identical results to the unsplit stream (the multi-channel front's chunk unit)
int fft_block_advance(int T, int D)
{
    int F = 1, ovlr = 0;
    if (fft_odd_tail(T, D, &F, nullptr, &ovlr))
        return F * (1024 - 64 * ovlr);
    return fft_two_partitions(T) ? FFT_N - FFT_PART : FFT_N - 64 * fft_overlap_rows(T, D);
}

#endif
template <int ROWS>
hipError_t launch_fft_rows(const LaunchArgs &a); // defined and explicitly instantiated in the unit compiled with IF_FIR_FFT_ROWS = ROWS
hipError_t launch_fft_two_partitions(const LaunchArgs &a); // (in the 32-row unit)
#ifdef IF_FIR_FFT_ROWS
template <int ROWS>
static hipError_t launch_all_slots(const LaunchArgs &p, bool nco) // the decimation-8 bank's all-slots form (tail 9)
{
    if constexpr (ROWS >= 4)
        switch ((p.in_i16 ? 2 : 0) | (nco ? 1 : 0))
        {
        case 0: return launch_fft_t<ROWS, true, false, false, 9>(p);
        case 1: return launch_fft_t<ROWS, true, false, true, 9>(p);
        case 2: return launch_fft_t<ROWS, true, true, false, 9>(p);
        default: return launch_fft_t<ROWS, true, true, true, 9>(p);
        }
    return hipErrorInvalidConfiguration;
}
template <int ROWS>
hipError_t launch_fft_rows(const LaunchArgs &a)
{
    int F = 1;
    const bool tail = fft_tail(a.T, a.D, &F, nullptr);
    const int key = (a.in_i16 ? 2 : 0) | (a.nco_word ? 1 : 0);
    if constexpr (ROWS >= 4)
    {
    if (a.chan)
    {
        // the filter bank: decimation 4, 8, 16; channels at their own centres (chan->general) also at every other multiple of 4 up to 64,
        // behind the tail of the largest of 16, 8, 4 that divides the decimation, keeping every sub-th output
        const int Fb = fft_bank_tail(a.D, a.chan->general != 0);
        if (!Fb || a.chan->count < 1 || a.chan->count > CHAN_MAX || (Fb == 4 && a.nco_word))
            return hipErrorInvalidConfiguration;
        const int ckey = (a.in_i16 ? 2 : 0) | (a.nco_word ? 1 : 0);
        if (Fb == 8) // per channel (pairs share a small inverse)
        {
            // slot form (chan->tw[] = W16^(a slot), a = 1..7) when every channel sits on the fs/16 grid and the context has no NCO;
            // the general form (chan->bin[] / pword[]: centre bin and mix-down word of a channel) otherwise
            bool general = a.chan->general || a.D != 8;
            for (uint32_t c = 0; c < a.chan->count; c++)
                general = general || (a.chan->bin[c] & 255u) || a.chan->pword[c] != (a.chan->bin[c] << 20) + a.nco_word;
            if (general)
                return a.in_i16 ? launch_fft_t<ROWS, true, true, true, 8>(a) : launch_fft_t<ROWS, true, false, true, 8>(a);
            const bool nco = a.nco_word != 0; // channels on the slot grid shifted by the context's NCO (a common offset)
            // Channels on the slot grid.  A parity (even / odd slots) with at least four channels, none listed twice, runs the
            // ALL-SLOTS form (round 4): one launch computes the eight slots of that parity from two 8-point transforms per group
            // (2340 packed instructions a block whatever the count, against 1008 + 415 per channel) and stores the wanted ones; the
            // other channels keep the per-channel form.  Both parities qualifying: ONE launch over virtual blocks (kernel).  Up to two
            // launches per call on the context's stream; only the first
            // one writes the next call's history.  (Even slots: the bank's own table image; odd slots: fft_tables_b.)
            const ChanArgs &cin = *a.chan;
            uint32_t pmask[2], rest = 0;
            // (diag 4096, development: per-channel form only)
            fft_bank8_plan(cin.slot, cin.count, a.fft_tables_b != nullptr && !(a.diag & 4096), pmask, &rest);
            bool first = true;
            if (pmask[0] && pmask[1] && !(a.diag & 8192)) // both parities: ONE launch over virtual blocks (diag 8192, development: two launches)
            {
                ChanArgs cs{};
                cs.count = (uint32_t)__builtin_popcount(pmask[0] | pmask[1]);
                cs.sub = 2u;
                cs.rot_e = cin.abs0n0 & 15u;
                cs.abs0n0 = cin.abs0n0;
                cs.mask16 = pmask[0] | pmask[1];
                for (uint32_t c = 0; c < cin.count; c++)
                    if ((cs.mask16 >> (cin.slot[c] & 15u)) & 1u)
                        cs.out[cin.slot[c] & 15u] = cin.out[c];
                LaunchArgs p = a;
                p.chan = &cs;
                const hipError_t e = launch_all_slots<ROWS>(p, nco);
                if (e != hipSuccess)
                    return e;
                first = false;
                pmask[0] = pmask[1] = 0;
            }
            for (uint32_t par = 0; par < 2; par++)
            {
                if (!pmask[par])
                    continue;
                ChanArgs cs{};
                cs.count = (uint32_t)__builtin_popcount(pmask[par]);
                cs.sub = par;
                cs.rot_e = cin.abs0n0 & 15u;
                cs.abs0n0 = cin.abs0n0;
                cs.mask16 = pmask[par];
                for (uint32_t c = 0; c < cin.count; c++)
                    if ((pmask[par] >> (cin.slot[c] & 15u)) & 1u)
                        cs.out[cin.slot[c] & 15u] = cin.out[c];
                LaunchArgs p = a;
                p.chan = &cs;
                p.fft_tables = par ? a.fft_tables_b : a.fft_tables; // (even slots: the bank's own image; odd slots: the image behind it)
                if (!first)
                    p.hist_out = nullptr;
                const hipError_t e = launch_all_slots<ROWS>(p, nco);
                if (e != hipSuccess)
                    return e;
                first = false;
            }
            ChanArgs cl{};
            for (uint32_t c = 0; c < cin.count; c++)
            {
                if (!((rest >> c) & 1u))
                    continue;
                const uint32_t k = cl.count++;
                cl.slot[k] = cin.slot[c];
                for (int w = 0; w < 30; w++)
                    cl.tw[k][w] = cin.tw[c][w];
                cl.rot0[k][0] = cin.rot0[c][0];
                cl.rot0[k][1] = cin.rot0[c][1];
                cl.out[k] = cin.out[c];
                cl.bin[k] = cin.bin[c];
                cl.pword[k] = cin.pword[c];
            }
            cl.abs0n0 = cin.abs0n0;
            if (!cl.count)
                return hipSuccess;
            LaunchArgs p = a;
            p.chan = &cl;
            if (!first)
                p.hist_out = nullptr;
            if (nco) // (the slot form proper has no NCO: the left-over channels of a shifted grid take the general form)
                return a.in_i16 ? launch_fft_t<ROWS, true, true, true, 8>(p) : launch_fft_t<ROWS, true, false, true, 8>(p);
            return a.in_i16 ? launch_fft_t<ROWS, true, true, false, 8>(p) : launch_fft_t<ROWS, true, false, false, 8>(p);
        }
        if (Fb == 16 && a.chan->general) // every channel at its own centre (per channel; arrays indexed by channel)
            return a.in_i16 ? launch_fft_t<ROWS, true, true, false, 17>(a) : launch_fft_t<ROWS, true, false, false, 17>(a);
        if (Fb == 16) // all 16 slots from one forward transform; chan->out[] / rot0[] are indexed by SLOT
            switch (ckey)
            {
            case 0: return launch_fft_t<ROWS, true, false, false, 16>(a);
            case 1: return launch_fft_t<ROWS, true, false, true, 16>(a);
            case 2: return launch_fft_t<ROWS, true, true, false, 16>(a);
            default: return launch_fft_t<ROWS, true, true, true, 16>(a);
            }
        if (a.chan->general && a.D == 4) // decimation 4, every channel at its own centre
            return a.in_i16 ? launch_fft_t<ROWS, true, true, false, 5>(a) : launch_fft_t<ROWS, true, false, false, 5>(a);
        if (a.chan->general) // decimation 12, 20, 28, ...: the same tail keeping every (D / 4)-th output
            return a.in_i16 ? launch_fft_t<ROWS, true, true, false, 6>(a) : launch_fft_t<ROWS, true, false, false, 6>(a);
        return a.in_i16 ? launch_fft_t<ROWS, true, true, false, 4>(a) : launch_fft_t<ROWS, true, false, false, 4>(a);
    }
    if (a.D == 2 && !a.no_fold) // frequency-domain fold + 2048-point inverse (round 3)
        switch (key)
        {
        case 0: return launch_fft_t<ROWS, true, false, false, 2>(a);
        case 1: return launch_fft_t<ROWS, true, false, true, 2>(a);
        case 2: return launch_fft_t<ROWS, true, true, false, 2>(a);
        default: return launch_fft_t<ROWS, true, true, true, 2>(a);
        }
    if (a.D == 4)
        switch (key)
        {
        case 0: return launch_fft_t<ROWS, true, false, false>(a);
        case 1: return launch_fft_t<ROWS, true, false, true>(a);
        case 2: return launch_fft_t<ROWS, true, true, false>(a);
        default: return launch_fft_t<ROWS, true, true, true>(a);
        }
    if (tail && F == 2 && !a.no_fold) // decimation 6, 10, ..., 62: the decimate-by-2 tail keeping every sub-th output
        switch (key)
        {
        case 0: return launch_fft_t<ROWS, true, false, false, 3>(a);
        case 1: return launch_fft_t<ROWS, true, false, true, 3>(a);
        case 2: return launch_fft_t<ROWS, true, true, false, 3>(a);
        default: return launch_fft_t<ROWS, true, true, true, 3>(a);
        }
    if (tail && F == 4) // decimation 8, 12, ..., 64: the decimate-by-4 tail keeping every sub-th output (tables as for decimation 4)
        switch (key)
        {
        case 0: return launch_fft_t<ROWS, true, false, false, 1>(a);
        case 1: return launch_fft_t<ROWS, true, false, true, 1>(a);
        case 2: return launch_fft_t<ROWS, true, true, false, 1>(a);
        default: return launch_fft_t<ROWS, true, true, true, 1>(a);
        }
    }
    else if (a.chan || (tail && !a.no_fold))
        return hipErrorInvalidConfiguration; // 2 overlap rows: full-rate pipeline only (fft_overlap_rows never sends a tail here)
    if (a.D == 1)
        switch (key)
        {
        case 0: return launch_fft_t<ROWS, false, false, false>(a);
        case 1: return launch_fft_t<ROWS, false, false, true>(a);
        case 2: return launch_fft_t<ROWS, false, true, false>(a);
        default: return launch_fft_t<ROWS, false, true, true>(a);
        }
    switch (key) // any other decimation: full-rate kernel + selecting store
    {
    case 0: return launch_fft_t<ROWS, false, false, false, false, true>(a);
    case 1: return launch_fft_t<ROWS, false, false, true, false, true>(a);
    case 2: return launch_fft_t<ROWS, false, true, false, false, true>(a);
    default: return launch_fft_t<ROWS, false, true, true, false, true>(a);
    }
}

#ifdef IF_FIR_FFT_ONLY // (development: ONE instantiation, e.g. -DIF_FIR_FFT_ONLY='4,true,false,false,17,false,false', to read its code)
__attribute__((used)) static auto *const if_fir_fft_only_kernel = &fir_fft_kernel<IF_FIR_FFT_ONLY>;
#elif defined(IF_FIR_FFT_HAZARD_PROBE) // (tests/test_host.py: one instantiation, the decimate-by-2 tail with its 16-byte stores)
template __global__ void fir_fft_kernel<IF_FIR_FFT_ROWS, true, false, false, 2, false, false>(
    const f2v *, f2v *, const f2v *, const f2v *, int, int64_t, int32_t, int64_t, int64_t, int64_t, unsigned int *,
    unsigned long long *, int32_t, uint32_t, uint32_t, chan_arg_t<2>, uint32_t, void *, int32_t, int32_t, int64_t, int32_t);
#else
template hipError_t launch_fft_rows<IF_FIR_FFT_ROWS>(const LaunchArgs &a);
#endif

#if IF_FIR_FFT_ROWS == 32 && !defined(IF_FIR_FFT_HAZARD_PROBE)
// Filters of 3074..4096 taps: h = (h_a, h_b) with 2048 taps in h_a.  Launch 1: y = h_a * x (writes the history);
// launch 2: y += h_b * x(n - 2048): the same kernel with h_b's table, reading the input FFT_PART samples late and adding
// its result to what launch 1 stored (ACC).  Both are the 32-row (2049-tap) kernel; an even decimation runs behind the
// decimating tails like shorter filters do (round 3), an odd one through the selecting store.  The history holds 4096
// samples: 2048 of delay + the overlap.
template <bool ACC>
static hipError_t launch_fft_partition(const LaunchArgs &a)
{
    int F = 1;
    (void)fft_tail(a.T, a.D, &F, nullptr); // (a.T: the whole filter's tap count)
    const int key = (a.in_i16 ? 2 : 0) | (a.nco_word ? 1 : 0);
#define IF_FIR_PART_SWITCH(DEC4, CHAN, DECN)                                            \
    switch (key)                                                                       \
    {                                                                                  \
    case 0: return launch_fft_t<32, DEC4, false, false, CHAN, DECN, ACC>(a);           \
    case 1: return launch_fft_t<32, DEC4, false, true, CHAN, DECN, ACC>(a);            \
    case 2: return launch_fft_t<32, DEC4, true, false, CHAN, DECN, ACC>(a);            \
    default: return launch_fft_t<32, DEC4, true, true, CHAN, DECN, ACC>(a);            \
    }
    if (a.D == 1)
        IF_FIR_PART_SWITCH(false, 0, false)
    if (F == 4 && a.D == 4)
        IF_FIR_PART_SWITCH(true, 0, false)
    if (F == 4)
        IF_FIR_PART_SWITCH(true, 1, false)
    if (F == 2 && a.D == 2 && !a.no_fold)
        IF_FIR_PART_SWITCH(true, 2, false)
    if (F == 2 && !a.no_fold)
        IF_FIR_PART_SWITCH(true, 3, false)
    IF_FIR_PART_SWITCH(false, 0, true)
#undef IF_FIR_PART_SWITCH
}

hipError_t launch_fft_two_partitions(const LaunchArgs &a)
{
    if (a.chan || !a.fft_tables_b || a.hist_len < 2 * FFT_PART)
        return hipErrorInvalidConfiguration;
    LaunchArgs p = a;
    const hipError_t e = launch_fft_partition<false>(p);
    if (e != hipSuccess)
        return e;
    p.fft_tables = a.fft_tables_b;
    p.in_shift = FFT_PART;
    p.hist_out = nullptr; // the first launch wrote the next history
    return launch_fft_partition<true>(p);
}
#endif // 32-row unit
#endif // IF_FIR_FFT_ROWS

#ifdef IF_FIR_FFT_HOST
hipError_t launch_fft(const LaunchArgs &a)
{
    if (!fft_supported(a.T, a.D) || !a.fft_tables)
        return hipErrorInvalidConfiguration;
    if (!a.chan && !a.no_fold && fft_odd_tail(a.T, a.D, nullptr, nullptr, nullptr))
        return launch_fft_odd(a);
    if (fft_two_partitions(a.T))
        return launch_fft_two_partitions(a);
    const int rows = fft_overlap_rows(a.T, a.D);
    switch (rows)
    {
    case 4: return launch_fft_rows<4>(a);
    case 8: return launch_fft_rows<8>(a);
    case 16: return launch_fft_rows<16>(a);
    case 32: return launch_fft_rows<32>(a);
    default: return launch_fft_rows<48>(a);
    }
}

// (cos, tan) form of the twiddle exp(j th): (c, t) with c = cos th rounded to float32 and t = sin th / c; an exact zero of the
// cosine is stored as +-2^-30 (its own contribution is below rounding, the tangent stays finite).  c_ref != 0: the first
// component is c / c_ref instead (third input of a radix-4 butterfly, see bfly4_tw).
static double tan_cos(double th)
{
    double c = cos(th);
    if (fabs(c) < 9.3e-10)
        c = (c < 0.0 ? -1.0 : 1.0) * 9.313225746154785e-10;
    return (double)(float)c;
}
static void tan_entry(double th, double c_ref, float *e)
{
    const double c = tan_cos(th);
    e[1] = (float)(sin(th) / c);
    e[0] = (float)(c_ref != 0.0 ? c / c_ref : c);
}
// the 15 table entries of fft16_tw for the base twiddle exp(j th): entry k at out[2 * k * stride]
static void tan_fft16_entries(double th, float *out, int stride)
{
    const double w16 = -6.283185307179586476925286766559 / 16.0;
    tan_entry(4.0 * th, 0.0, out);
    tan_entry(8.0 * th, 0.0, out + 2 * stride);
    tan_entry(12.0 * th, tan_cos(4.0 * th), out + 4 * stride);
    for (int q = 0; q < 4; q++)
    {
        const double b = th + w16 * q;
        tan_entry(b, 0.0, out + 2 * (3 + 3 * q) * stride);
        tan_entry(2.0 * b, 0.0, out + 2 * (4 + 3 * q) * stride);
        tan_entry(3.0 * b, tan_cos(b), out + 2 * (5 + 3 * q) * stride);
    }
}

// the filter banks' forward passes 2 and 3 (first stage) in (cos, tan) form, as in the decimate-by-4 image (fft_build_tables)
// the filter-bank images' table between the two transforms of the small inverse (inverse_tail256_tan): twet[e * 16 + mu1], b = W256^mu1
static void bank_tan_inverse(float *twe)
{
    const double PI2 = 6.283185307179586476925286766559;
    for (int e = 0; e < 2 * 1024; e++)
        twe[e] = 0.0f;
    for (int mu1 = 0; mu1 < 16; mu1++)
        tan_fft16_entries(-PI2 * (double)mu1 / 256.0, twe + 2 * mu1, 16);
}

static void bank_tan_forward(float *tw1, float *tw2)
{
    const double PI2 = 6.283185307179586476925286766559;
    for (int e = 0; e < 2 * 4096; e++)
        tw1[e] = 0.0f;
    for (int e = 0; e < 2 * 256; e++)
        tw2[e] = 0.0f;
    for (int i = 0; i < 4; i++)
    {
        for (int lane = 0; lane < 64; lane++)
        {
            float all[30];
            tan_fft16_entries(-PI2 * (double)((4 * (lane / 16) + i) + 16 * (lane % 16)) / 4096.0, all, 1);
            for (int e = 0; e < 3; e++)
            {
                tw1[2 * ((i * 3 + e) * 64 + lane) + 0] = all[2 * e];
                tw1[2 * ((i * 3 + e) * 64 + lane) + 1] = all[2 * e + 1];
            }
        }
        for (int g = 0; g < 4; g++)
            tan_fft16_entries(-PI2 * (double)(4 * g + i) / 256.0, tw2 + 2 * (i * 60 + g), 4);
    }
}

// the phasor tables of lds_phasor, at bytes [6 KB, 8 KB) of the image: P1[k] = exp(j 2 pi k / 128), P2[k] = exp(j 2 pi k / 16384)
void fft_phasor_tables(float *tables)
{
    const double PI2 = 6.283185307179586476925286766559;
    float *pht = tables + 6144 / 4;
    for (int k = 0; k < 128; k++)
    {
        pht[2 * k + 0] = (float)cos(PI2 * (double)k / 128.0);
        pht[2 * k + 1] = (float)sin(PI2 * (double)k / 128.0);
        pht[2 * (128 + k) + 0] = (float)cos(PI2 * (double)k / 16384.0);
        pht[2 * (128 + k) + 1] = (float)sin(PI2 * (double)k / 16384.0);
    }
}

// Host side: twiddle and H tables in the kernel's LDS image order (float64 math, rounded once to float32).
//   [0, 32 KB)      tw1[(rho*16+k0)*64 + lane] = W4096^((lane+64*rho)*k0)
//   [32 KB, 64 KB)  hp [(i*16+k2)*64 + lane]   = FFT(taps)[(4*(lane/16)+i) + 16*(lane%16) + 256*k2] / 4096
//   [64 KB, 66 KB)  tw2[k1*16 + n2]            = W256^(n2*k1)
//   [66 KB, 82 KB)  twd, twe: twiddles of the decimate-by-4 1024-point inverse
void fft_build_tables(const float *taps, int T, int ctaps, int D, uint32_t nco_delta, double in_scale,
                      float *tables /* FFT_TABLE_FLOATS floats */, int bank, int full_rate, int bank_parity)
{
    const double PI2 = 6.283185307179586476925286766559;
    float *tw1 = tables, *hp = tables + 2 * 4096, *tw2 = tables + 4 * 4096;
    float *twd = tw2 + 2 * 256, *twe = twd + 2 * 1024, *ncob = twe + 2 * 1024, *twf = ncob + 2 * 64;
    // decimate-by-2 inverse: twf[i*64 + lane] = W2048^(16*(lane%16) + 4*(lane/16) + i)
    for (int i = 0; i < 4; i++)
        for (int lane = 0; lane < 64; lane++)
        {
            const int e = 16 * (lane % 16) + 4 * (lane / 16) + i;
            const double a = -PI2 * (double)e / 2048.0;
            twf[2 * (i * 64 + lane) + 0] = (float)cos(a);
            twf[2 * (i * 64 + lane) + 1] = (float)sin(a);
        }
    // NCO (SPEC §3.2): rotation shared by the 64 outputs of row r of a block, exp(+j*2*pi*((64 r delta) mod 2^32)/2^32)
    // (the decimating filter-bank tails store 16 / 32 outputs per slot: bank = 16 / 8)
    const uint32_t nco_step = bank == 16 ? 16u : bank == 8 ? 32u : 64u;
    for (uint32_t r = 0; r < 64; r++)
    {
        // (bank tails: entries 32..63 hold the lane's share, phasor((r - 32) delta): output index within a slot; entries
        // 16..31 the 16th roots of unity W16^(r - 16) = exp(-j 2 pi (r - 16) / 16): the slots' mix-down phases)
        const uint32_t ph = (bank && r >= 32) ? (r - 32u) * nco_delta
                            : (bank && r >= 16) ? 0u - ((r - 16u) << 28) : nco_step * r * nco_delta;
        const double a = PI2 * ((double)ph / 4294967296.0);
        ncob[2 * r + 0] = (float)cos(a);
        ncob[2 * r + 1] = (float)sin(a);
    }
    // decimate-by-4 inverse: twd[(i*4+mu2)*64 + lane] = W1024^((16*(lane%16) + 4*(lane/16) + i)*mu2)
    //                        twe[mu1*64 + lane]       = W256^((4*(lane/16) + (lane%16)/4)*mu1)
    for (int i = 0; i < 4; i++)
        for (int mu2 = 0; mu2 < 4; mu2++)
            for (int lane = 0; lane < 64; lane++)
            {
                const int e = ((16 * (lane % 16) + 4 * (lane / 16) + i) * mu2) % 1024;
                const double a = -PI2 * (double)e / 1024.0;
                twd[2 * ((i * 4 + mu2) * 64 + lane) + 0] = (float)cos(a);
                twd[2 * ((i * 4 + mu2) * 64 + lane) + 1] = (float)sin(a);
            }
    for (int mu1 = 0; mu1 < 16; mu1++)
        for (int lane = 0; lane < 64; lane++)
        {
            const int e = ((4 * (lane / 16) + (lane % 16) / 4) * mu1) % 256;
            const double a = -PI2 * (double)e / 256.0;
            twe[2 * (mu1 * 64 + lane) + 0] = (float)cos(a);
            twe[2 * (mu1 * 64 + lane) + 1] = (float)sin(a);
        }
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
    std::vector<double> ct(4096), st(4096), hd(2 * 4096); // not static: contexts may be created from several threads
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
                    const double hr = ctaps ? (double)taps[2 * n] : (double)taps[n];
                    const double hi = ctaps ? (double)taps[2 * n + 1] : 0.0;
                    re += hr * ct[e] - hi * st[e];
                    im += hr * st[e] + hi * ct[e];
                }
                hd[2 * ((i * 16 + k2) * 64 + lane) + 0] = re / 4096.0 * in_scale; // in_scale: 2^-15 for raw int16 samples
                hd[2 * ((i * 16 + k2) * 64 + lane) + 1] = im / 4096.0 * in_scale;
            }
    if (bank == 8)
    {
        // filter bank at decimation 8: G_q[a] = W16^(a q) sum_j H(k0 + 16 k1 + 256 (q + 2 j)) W8^(a j) at ((i*16 + 8 q + a)*64 + lane)
        // (second stage of pass 3, the multiplication by the slot's H and the 8-way alias fold merged; kernel, CHAN == 8)
        for (int i = 0; i < 4; i++)
            for (int lane = 0; lane < 64; lane++)
                for (int q = 0; q < 2; q++)
                    for (int a8 = 0; a8 < 8; a8++)
                    {
                        double re = 0.0, im = 0.0;
                        for (int j = 0; j < 8; j++)
                        {
                            const double *h = &hd[2 * ((i * 16 + q + 2 * j) * 64 + lane)];
                            const int e = (256 * a8 * q + 512 * a8 * j) & 4095;
                            re += h[0] * ct[e] - h[1] * st[e];
                            im += h[0] * st[e] + h[1] * ct[e];
                        }
                        // Round 4: the bank's forward passes are the decimate-by-4 kernels' ((cos, tan) twiddles on the inputs of passes 2
                        // and 3), so the factor b^a, b = W4096^(k0 + 16 k1), that input a of pass 3 still carries is folded in here.
                        // This image serves the per-channel forms (CHAN == 8) and the all-slots form's even slots (CHAN == 9);
                        // bank_parity = 1: the all-slots form's image for the ODD slots -- also the slot twiddle's common factor W16^a,
                        // and the halves exchanged (first half: the factor of w0 = G_1, second: G_0)
                        int half = q;
                        {
                            const int par = bank_parity ? 1 : 0;
#if IF_FIR_FFT_TAN
                            const int eb = a8 * ((4 * (lane / 16) + i) + 16 * (lane % 16)); // b^a
#else
                            const int eb = 0;
#endif
                            const int e = (256 * a8 * par + eb) & 4095; // W16^(a par) b^a
                            const double gr = re * ct[e] - im * st[e], gi = re * st[e] + im * ct[e];
                            re = gr;
                            im = gi;
                            half = par ? 1 - q : q;
                        }
                        hp[2 * ((i * 16 + 8 * half + a8) * 64 + lane) + 0] = (float)re;
                        hp[2 * ((i * 16 + 8 * half + a8) * 64 + lane) + 1] = (float)im;
                    }
#if IF_FIR_FFT_TAN
        bank_tan_forward(tw1, tw2); // (the 512-point inverse keeps twd above)
        bank_tan_inverse(twe);
        fft_phasor_tables(tables);
#endif
        return;
    }
    if (bank == 16)
    {
        // 16-slot filter bank at the channel rate: G0[n2] = sum_k2 H(k0 + 16 k1 + 256 k2) W16^(n2 k2) at ((i*16 + n2)*64 + lane)
        // (pass 3, the multiplication by the slot's H and the 16-way alias fold merged; see the kernel, CHAN == 16)
        for (int i = 0; i < 4; i++)
            for (int lane = 0; lane < 64; lane++)
                for (int n2 = 0; n2 < 16; n2++)
                {
                    double re = 0.0, im = 0.0;
                    for (int k2 = 0; k2 < 16; k2++)
                    {
                        const double *h = &hd[2 * ((i * 16 + k2) * 64 + lane)];
                        const int e = (256 * n2 * k2) & 4095;
                        re += h[0] * ct[e] - h[1] * st[e];
                        im += h[0] * st[e] + h[1] * ct[e];
                    }
#if IF_FIR_FFT_TAN
                    {
                        // (round 4) the b^n2, b = W4096^(k0 + 16 k1), that input n2 of pass 3 still carries ((cos, tan) forward passes)
                        const int eb = (n2 * ((4 * (lane / 16) + i) + 16 * (lane % 16))) & 4095;
                        const double gr = re * ct[eb] - im * st[eb], gi = re * st[eb] + im * ct[eb];
                        re = gr;
                        im = gi;
                    }
#endif
                    hp[2 * ((i * 16 + n2) * 64 + lane) + 0] = (float)re;
                    hp[2 * ((i * 16 + n2) * 64 + lane) + 1] = (float)im;
                }
#if IF_FIR_FFT_TAN
        bank_tan_forward(tw1, tw2);
        bank_tan_inverse(twe);
        fft_phasor_tables(tables);
#endif
        return;
    }
    if (D != 4)
    {
        for (int e = 0; e < 2 * 4096; e++)
            hp[e] = (float)hd[e];
#if IF_FIR_FFT_TAN
        if (full_rate)
        {
            // the full-rate pipeline's twiddles in (cos, tan) form (LDS map at the top of the file): forward pass 2 and the first
            // stage of forward pass 3 as in the decimate-by-4 image; the shared table T; inverse pass 2
            for (int e = 0; e < 2 * 4096; e++)
                tw1[e] = 0.0f;
            for (int e = 0; e < 2 * 256; e++)
                tw2[e] = 0.0f;
            for (int e = 0; e < 2 * 1024; e++)
                twd[e] = 0.0f;
            for (int i = 0; i < 4; i++)
            {
                for (int lane = 0; lane < 64; lane++)
                {
                    float all[30];
                    tan_fft16_entries(-PI2 * (double)((4 * (lane / 16) + i) + 16 * (lane % 16)) / 4096.0, all, 1);
                    for (int e = 0; e < 3; e++)
                    {
                        tw1[2 * ((i * 3 + e) * 64 + lane) + 0] = all[2 * e];
                        tw1[2 * ((i * 3 + e) * 64 + lane) + 1] = all[2 * e + 1];
                    }
                }
                for (int g = 0; g < 4; g++)
                    tan_fft16_entries(-PI2 * (double)(4 * g + i) / 256.0, tw2 + 2 * (i * 60 + g), 4);
            }
            float *tt = tw1 + 2 * 1024; // (LDS_TT)
            for (int m = 0; m < 1024; m++)
            {
                const double th = -PI2 * (double)m / 4096.0;
                const unsigned pos = tsw((unsigned)m);
                tan_entry(th, 0.0, tt + 2 * pos);
                tan_entry(2.0 * th, 0.0, tt + 2 * (1024 + pos));
                tan_entry(3.0 * th, tan_cos(th), tt + 2 * (2048 + pos));
            }
            for (int n2 = 0; n2 < 16; n2++)
                tan_fft16_entries(-PI2 * (double)n2 / 256.0, twd + 2 * n2, 16);
            fft_phasor_tables(tables);
        }
#else
        (void)full_rate;
#endif
        return;
    }
    // decimate-by-4 kernels: the table holds G[m0][q] = W16^(m0 q) * sum_p H(q + 4p) W4^(m0 p) at ((i*16 + 4*m0 + q)*64 + lane)
    // (second radix-4 stage of pass 3, multiplication by H and alias fold merged; see the kernel)
    for (int i = 0; i < 4; i++)
        for (int lane = 0; lane < 64; lane++)
            for (int m0 = 0; m0 < 4; m0++)
                for (int q = 0; q < 4; q++)
                {
                    double re = 0.0, im = 0.0;
                    for (int p = 0; p < 4; p++)
                    {
                        const double *h = &hd[2 * ((i * 16 + q + 4 * p) * 64 + lane)];
                        const int e = (256 * m0 * q + 1024 * m0 * p) & 4095; // W16^(m0 q) W4^(m0 p) as a power of W4096
                        re += h[0] * ct[e] - h[1] * st[e];
                        im += h[0] * st[e] + h[1] * ct[e];
                    }
#if IF_FIR_FFT_TAN
                    {
                        // G' = b^m0 G, b = W4096^(k0 + 16 k1): the factor the first stage of pass 3 still owes (see the kernel)
                        const int eb = (m0 * ((4 * (lane / 16) + i) + 16 * (lane % 16))) & 4095;
                        const double gr = re * ct[eb] - im * st[eb], gi = re * st[eb] + im * ct[eb];
                        re = gr;
                        im = gi;
                    }
#endif
                    hp[2 * ((i * 16 + 4 * m0 + q) * 64 + lane) + 0] = (float)re;
                    hp[2 * ((i * 16 + 4 * m0 + q) * 64 + lane) + 1] = (float)im;
                }
#if IF_FIR_FFT_TAN
    // the decimate-by-4 kernels' twiddles in (cos, tan) form, in the slots of the tables they replace (LDS map at the top)
    for (int e = 0; e < 2 * 4096; e++)
        tw1[e] = 0.0f;
    for (int e = 0; e < 2 * 256; e++)
        tw2[e] = 0.0f;
    for (int e = 0; e < 2 * 1024; e++)
        twd[e] = twe[e] = 0.0f;
    for (int i = 0; i < 4; i++)
    {
        for (int lane = 0; lane < 64; lane++)
        {
            // pass 3, first stage: b = W4096^(k0 + 16 k1); only entries 0..2 (b^4, b^8, b^12) are used
            float all[30];
            tan_fft16_entries(-PI2 * (double)((4 * (lane / 16) + i) + 16 * (lane % 16)) / 4096.0, all, 1);
            for (int e = 0; e < 3; e++)
            {
                tw1[2 * ((i * 3 + e) * 64 + lane) + 0] = all[2 * e];
                tw1[2 * ((i * 3 + e) * 64 + lane) + 1] = all[2 * e + 1];
            }
        }
        for (int g = 0; g < 4; g++) // pass 2: b = W256^k0, k0 = 4 g + i
            tan_fft16_entries(-PI2 * (double)(4 * g + i) / 256.0, tw2 + 2 * (i * 60 + g), 4);
    }
    for (int lane = 0; lane < 64; lane++) // inverse, last pass: b = W1024^lane
        tan_fft16_entries(-PI2 * (double)lane / 1024.0, twd + 2 * lane, 64);
    for (int mu2 = 0; mu2 < 4; mu2++) // inverse, middle pass: b = W64^mu2
        tan_fft16_entries(-PI2 * (double)mu2 / 64.0, twe + 2 * mu2, 4);
    fft_phasor_tables(tables);
#endif
}

// Table image of the odd-decimation kernel (fir_odd_kernel, F = 3 or 5), float64 math, rounded once:
//   G_p [(p*16 + slot)*64 + lane] = FFT1024(g_p)[k0 + 16 k1 + 256 k2'] / 1024 (slot = 4 i + k2', k0 = 4 (lane/16) + i, k1 = lane%16),
//        g_0[k] = h[F k], g_p[d] = h[F d - p] (p >= 1, d >= 1);  in_scale: 2^-15 for raw int16 samples
//   TB [e*16 + k0] (b = W256^k0) | TC [(i*3 + e)*64 + lane] (b = W1024^(k0 + 16 k1), entries 0..2 of the transform's 15)
//   TWD [e*64 + lane] (b = W1024^lane) | TWE [e*4 + mu2] (b = W64^mu2) | NCO row phasors (64 outputs per row)
void fft_build_tables_odd(const float *taps, int T, int ctaps, int F, uint32_t nco_delta, double in_scale, float *tables)
{
    const double PI2 = 6.283185307179586476925286766559;
    float *gt = tables, *tb = gt + 2 * F * 1024, *tc = tb + 2 * 256, *twd = tc + 2 * 768, *twe = twd + 2 * 1024, *ncob = twe + 2 * 64;
    for (int e = 0; e < 2 * (256 + 768 + 1024 + 64 + 64); e++)
        tb[e] = 0.0f;
    for (int k0 = 0; k0 < 16; k0++)
        tan_fft16_entries(-PI2 * (double)k0 / 256.0, tb + 2 * k0, 16);
    for (int i = 0; i < 4; i++)
        for (int lane = 0; lane < 64; lane++)
        {
            float all[30];
            tan_fft16_entries(-PI2 * (double)((4 * (lane / 16) + i) + 16 * (lane % 16)) / 1024.0, all, 1);
            // (the last pass is a 4-point DFT over mu2: its inputs carry b, b^2, b^3 = entries 3, 4, 5 of the 16-point set, q = 0)
            for (int e = 0; e < 3; e++)
            {
                tc[2 * ((i * 3 + e) * 64 + lane) + 0] = all[2 * (3 + e)];
                tc[2 * ((i * 3 + e) * 64 + lane) + 1] = all[2 * (3 + e) + 1];
            }
        }
    for (int lane = 0; lane < 64; lane++)
        tan_fft16_entries(-PI2 * (double)lane / 1024.0, twd + 2 * lane, 64);
    for (int mu2 = 0; mu2 < 4; mu2++)
        tan_fft16_entries(-PI2 * (double)mu2 / 64.0, twe + 2 * mu2, 4);
    for (uint32_t r = 0; r < 64; r++)
    {
        const double a = PI2 * ((double)(uint32_t)(64u * r * nco_delta) / 4294967296.0);
        ncob[2 * r + 0] = (float)cos(a);
        ncob[2 * r + 1] = (float)sin(a);
    }
    std::vector<double> ct(1024), st(1024), gr(1024), gi(1024);
    for (int e = 0; e < 1024; e++)
    {
        ct[e] = cos(-PI2 * (double)e / 1024.0);
        st[e] = sin(-PI2 * (double)e / 1024.0);
    }
    for (int p = 0; p < F; p++)
    {
        // polyphase component p at the decimated rate, with the delay of its phase stream
        for (int e = 0; e < 1024; e++)
            gr[e] = gi[e] = 0.0;
        for (int k = 0; k < T; k++)
        {
            // x[F m - k] = x_p[m - d] with F d - p = k
            if ((k + p) % F)
                continue;
            const int d = (k + p) / F;
            gr[d] += ctaps ? (double)taps[2 * k] : (double)taps[k];
            gi[d] += ctaps ? (double)taps[2 * k + 1] : 0.0;
        }
        int dmax = 0;
        for (int e = 0; e < 1024; e++)
            if (gr[e] != 0.0 || gi[e] != 0.0)
                dmax = e;
        for (int slot = 0; slot < 16; slot++)
            for (int lane = 0; lane < 64; lane++)
            {
                const int k = (4 * (lane / 16) + slot / 4) + 16 * (lane % 16) + 256 * (slot % 4);
                double re = 0.0, im = 0.0;
                for (int d = 0; d <= dmax; d++)
                {
                    const int e = (int)(((int64_t)k * d) & 1023);
                    re += gr[d] * ct[e] - gi[d] * st[e];
                    im += gr[d] * st[e] + gi[d] * ct[e];
                }
                gt[2 * ((p * 16 + slot) * 64 + lane) + 0] = (float)(re / 1024.0 * in_scale);
                gt[2 * ((p * 16 + slot) * 64 + lane) + 1] = (float)(im / 1024.0 * in_scale);
            }
    }
}

#endif // host side

} // namespace if_fir
