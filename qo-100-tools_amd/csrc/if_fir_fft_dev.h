// if_fir_fft_dev.h -- part of the overlap-save kernel's source (if_fir_fft.hip includes it; not a translation unit of its own).
// Shared device side: complex arithmetic in packed FP32, the 16-/8-point transforms, lane exchanges, memory helpers, block queue, the small inverses, gather_mac.
#pragma once
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
// IF_FIR_FFT_TW_PREFETCH (round 5 experiment, profiles/r05_table_prefetch_ab.txt): 1 = all 15 entries are requested at the top and a
// scheduling barrier keeps them there (the scheduler otherwise sinks every table read down to its use: one LDS round trip per
// butterfly, waited for on the spot); 0 = as written, the compiler places them
#ifndef IF_FIR_FFT_TW_PREFETCH
#define IF_FIR_FFT_TW_PREFETCH 1
#endif
// N table entries p[k STRIDE] requested together, and a scheduling barrier that keeps the requests here (round 5: see fft16_tw)
template <int N, int STRIDE, bool PF = (IF_FIR_FFT_TW_PREFETCH != 0)>
__device__ __forceinline__ void lds_fetch(cf (&w)[N], const f2v *p)
{
#pragma unroll
    for (int k = 0; k < N; k++)
        w[k] = p[k * STRIDE];
    if constexpr (PF)
        __builtin_amdgcn_sched_barrier(0);
}
// PF (round 5, profiles/r05_table_prefetch_ab.txt): all 15 entries are requested at the top and a scheduling barrier keeps them
// there.  Left to itself the machine scheduler sinks every table read down to its use -- it minimises live registers -- and the
// wave waits one LDS round trip per butterfly, on the spot; with the reads up front the headline kernel runs 6 % faster on the same
// instructions.  30 registers for the duration of the transform: the tails that have none to spare pass PF = false.
template <bool INV, int STRIDE, bool PF = (IF_FIR_FFT_TW_PREFETCH != 0)>
__device__ __forceinline__ void fft16_tw(cf (&v)[16], const f2v *e)
{
    cf y[4][4];
    if constexpr (PF)
    {
        cf ee[15];
#pragma unroll
        for (int k = 0; k < 15; k++)
            ee[k] = e[k * STRIDE];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++)
            bfly4_tw<INV>(v[i], v[i + 4], v[i + 8], v[i + 12], ee[0], ee[1], ee[2], y[0][i], y[1][i], y[2][i], y[3][i]);
#pragma unroll
        for (int q = 0; q < 4; q++)
            bfly4_tw<INV>(y[q][0], y[q][1], y[q][2], y[q][3], ee[3 + 3 * q], ee[4 + 3 * q], ee[5 + 3 * q], v[q], v[q + 4], v[q + 8], v[q + 12]);
    }
    else
    {
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
}

// The same with the table in two pieces (round 4, full-rate pipeline): the first stage's entries at s1[0], s1[S1], s1[2 S1]; the
// second stage's from the SHARED table T of the triples (b, b^2, b^3 with the third as (c3 / c1, t3)) of b = W4096^m, m = 0..1023:
// entry (b W16^q)^(j+1) = T_j[m + 256 q] at tq[1024 j + 256 q] (three arrays of 1024 entries; tq = T + tsw(m)).
// (PF: the 15 entries requested at the top, kept there by a scheduling barrier -- see fft16_tw)
template <bool INV, int S1, bool PF = (IF_FIR_FFT_TW_PREFETCH != 0)>
__device__ __forceinline__ void fft16_tw_T(cf (&v)[16], const f2v *s1, const f2v *tq)
{
    cf y[4][4];
    if constexpr (PF)
    {
        cf ee[15];
        ee[0] = s1[0];
        ee[1] = s1[S1];
        ee[2] = s1[2 * S1];
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            ee[3 + 3 * q] = tq[256 * q];
            ee[4 + 3 * q] = tq[1024 + 256 * q];
            ee[5 + 3 * q] = tq[2048 + 256 * q];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++)
            bfly4_tw<INV>(v[i], v[i + 4], v[i + 8], v[i + 12], ee[0], ee[1], ee[2], y[0][i], y[1][i], y[2][i], y[3][i]);
#pragma unroll
        for (int q = 0; q < 4; q++)
            bfly4_tw<INV>(y[q][0], y[q][1], y[q][2], y[q][3], ee[3 + 3 * q], ee[4 + 3 * q], ee[5 + 3 * q], v[q], v[q + 4], v[q + 8], v[q + 12]);
    }
    else
    {
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
// (round 5) the X exchange's own region stride: with single ds_read_b64 (lanes 0-31 / 32-63 per LDS cycle, 64 banks) the reads of the lane rows
// g = 0, 1 (and 2, 3) go out together, and their 16 x 136-byte rows must interleave on the banks: region stride = 128 bytes mod 256 (with
// XREG's +32 the two rows collide on 8 of 64 banks: SQ_LDS_BANK_CONFLICT 12 % of the LDS cycles, profiles/r05_lds_single_reads.txt).  The Y
// exchange keeps XREG: its stores (16 lanes per LDS cycle, 32 banks) need the +32, and no stride serves both its stores and its reads.
#ifndef IF_FIR_FFT_XREGX_PAD
#define IF_FIR_FFT_XREGX_PAD 0
#endif
constexpr int XREGX = 16 * XROW + IF_FIR_FFT_XREGX_PAD;
static_assert(XREGX <= XREG, "the X exchange's regions fit the per-wave buffer");
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
    return XAddr{xb + g * XREGX + m * 8, lds_opaque(xb + g * XREGX + m * XROW), nullptr, nullptr};
}
__device__ __forceinline__ XAddr xaddr_xy(char *xb, int lane)
{
    const int g = lane >> 4, m = lane & 15;
    const int k0 = 4 * g + (m >> 2), low = m & 3;
    return XAddr{xb + g * XREGX + m * 8, lds_opaque(xb + g * XREGX + m * XROW), xb + low * XREG + k0 * 8,
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
// small inverse).  Round 2 measured 4 = 3 and kept 3; round 5, with the steady state's pass 1 no longer waiting for the previous
// block's stores (IF_FIR_FFT_COLD_WAIT) the fourth batch's extra lead is worth 0.1-0.5 % (profiles/r05_table_prefetch_ab.txt): 4.
#ifndef IF_FIR_FFT_COLD_WAIT
#define IF_FIR_FFT_COLD_WAIT 1 // the cold load path drains its loads before it joins the steady-state path (if_fir_fft_kernel.inc)
#endif
// (round 5, profiles/r05_lds_single_reads.txt) LDS reads as single ds_read_b64: the compiler's machine-level load/store optimizer pairs the
// kernels' 8-byte LDS reads into ds_read2_b64 / ds_read2st64_b64, which the LDS serves in 8 cycles per pair on 32 banks, where two ds_read_b64 take
// 2 cycles each on 64 banks (MI355X_MICROARCH.md, LDS); the LDS array was busy 60 % of the time in these kernels.  Per kernel, through the
// subtarget feature (device pass only: the host pass does not know the feature); the IR-level vectorizer, which merges ADJACENT pairs, is
// switched off for the units in csrc/Makefile (-mllvm -amdgpu-load-store-vectorizer=0).
#ifndef IF_FIR_FFT_SINGLE_READS
#define IF_FIR_FFT_SINGLE_READS 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && IF_FIR_FFT_SINGLE_READS
#define IF_FIR_LDS_SINGLE_READS __attribute__((target("no-load-store-opt")))
#else
#define IF_FIR_LDS_SINGLE_READS
#endif
// decimate-by-2 tail: its 16 entries of H per group in rolling batches of this many (0: read where they are used, round 4's form; 8 spills).
// With batches of 4 and the 4-point stage's twiddles of its small inverses requested ahead the tail runs on single LDS reads like the others:
// 255 taps /2 -2.6 %, int16 -4.3 %, with the NCO -5.4 % against round 4's form with paired reads (profiles/r05_lds_single_reads.txt, last section)
#ifndef IF_FIR_FFT_DEC2_PF
#define IF_FIR_FFT_DEC2_PF 4
#endif
// decimate-by-2 tail: which table reads of its two small inverses are requested ahead (bit 0: the 4-point stage's, bit 1: the 16-point stages')
#ifndef IF_FIR_FFT_DEC2_PFI
#define IF_FIR_FFT_DEC2_PFI 1 // (bit 1 spills)
#endif
#ifndef IF_FIR_ODD_YREG
#define IF_FIR_ODD_YREG 2240 // odd-decimation kernel: region stride of the forward transforms' Y^-1 (if_fir_fft_odd.inc)
#endif
#ifndef IF_FIR_FFT_EARLY_GROUPS
#define IF_FIR_FFT_EARLY_GROUPS 4
#endif
// (the tail that keeps every sub-th output, CHAN 1 = decimation 8, 12, ..., 64, computes 15 store offsets on top: with four early batches and
// unpaired LDS reads two of its instantiations' cold paths needed 4 VGPRs of scratch; three batches there)
#ifndef IF_FIR_FFT_EARLY_GROUPS_SUB
#define IF_FIR_FFT_EARLY_GROUPS_SUB 3
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
    // (u < 2^12, magic <= 2^17, qd sub <= u: 24-bit multiplies, full rate; the 32-bit ones run at a quarter of it)
    __device__ __forceinline__ int64_t index(unsigned off) const
    {
        const unsigned u = rem + off, qd = __umul24(u, magic) >> 18;
        return (__umul24(qd, sub) == u) ? qU + (int64_t)qd : (int64_t)-1;
    }
    // the same as the byte offset of an 8-byte store through a descriptor that starts at the block's first kept output, qb = qU + (rem ? 1 : 0):
    // (index - qb) * 8 = (qd - (rem ? 1 : 0)) * 8 in 32-bit arithmetic, or 0xffffffff (out of range = dropped) for an output that is not kept or
    // not `in_range` (round 5: the 64-bit form of this cost the tail that keeps every sub-th output 1.3 us of its 8.8 us block)
    __device__ __forceinline__ unsigned store_offset(unsigned off, bool in_range) const
    {
        const unsigned u = rem + off, qd = __umul24(u, magic) >> 18;
        const bool keep = (__umul24(qd, sub) == u) && in_range;
        return keep ? (qd - (rem ? 1u : 0u)) * 8u : 0xffffffffu;
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
template <bool PF = (IF_FIR_FFT_TW_PREFETCH != 0)>
__device__ __forceinline__ void inverse_tail256(cf (&a)[16], cf (&c)[16], const f2v *twe, const XAddr &xa, int lane)
{
    xst16<XROW>(xa.wx, a);
    xld16<8>(xa.rx, a);
    cf we[15];
    lds_fetch<15, 64, PF>(we, twe + 64 + lane); // (round 5: the 15 twiddles requested ahead of the transform)
    fft16<true>(a); // over k1 -> mu1
#pragma unroll
    for (int mu1 = 1; mu1 < 16; mu1++)
        a[mu1] = cmul_v<true>(a[mu1], we[mu1 - 1]);
    // Y: wy + mu1 XROW = element (mu1, k0) of region `low`; ry + 8 k0
    xst16<XROW>(xa.wy, a);
    xld16<8>(xa.ry, c);
    fft16<true>(c); // over k0 -> mu0
}

// The same for the filter-bank images (round 5): the twiddle conj W256^(k0 mu1) between the two transforms sits on the INPUTS of the
// second one in (cos, tan) form -- input k0 of lane 4 mu1 + low carries conj(b)^k0, b = W256^mu1 (table twet[e * 16 + mu1], the 15
// entries of fft16_tw) -- 88 packed instructions where the 15 multiplies + the plain transform are 110.
template <bool PF = (IF_FIR_FFT_TW_PREFETCH != 0)>
__device__ __forceinline__ void inverse_tail256_tan(cf (&a)[16], cf (&c)[16], const f2v *twet, const XAddr &xa, int lane)
{
    xst16<XROW>(xa.wx, a);
    xld16<8>(xa.rx, a);
    fft16<true>(a); // over k1 -> mu1
    xst16<XROW>(xa.wy, a);
    xld16<8>(xa.ry, c);
    fft16_tw<true, 16, PF>(c, twet + (lane >> 2)); // over k0 -> mu0
}

// decimate-by-4 tail of one block: the 4 spectral aliases are folded in-lane (k2 = k2' + 4j) and a 1024-point inverse
// (4 x 16 x 16, tools/fft_model.py inverse_dec4) produces y[4m'] directly: lane = 4*mu1+mu2, slot mu0 -> y_D[64*mu0+lane]
template <bool PF = (IF_FIR_FFT_TW_PREFETCH != 0), bool PFE = PF>
__device__ __forceinline__ void inverse_dec4(const cf (&z)[16], cf (&c)[16], const f2v *twd, const f2v *twe, const XAddr &xa,
                                             int lane)
{
    cf a[16], wd[16];
    lds_fetch<16, 64, PF>(wd, twd + lane); // (round 5: the twiddles requested ahead of the butterflies; entries 4 i are unused ones)
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        bfly4<true>(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3], a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
#pragma unroll
        for (int mu2 = 1; mu2 < 4; mu2++)
            a[4 * i + mu2] = cmul_v<true>(a[4 * i + mu2], wd[i * 4 + mu2]);
    }
    inverse_tail256<PFE>(a, c, twe, xa, lane);
}

// the same with the twiddles in (cos, tan) form on the inputs of the two 16-point transforms (round 4; tables tb = LDS_TWE,
// tc = LDS_TWD): 4-point inverse over k2' (plain) -> X -> iFFT16 over k1, inputs carry conj(W64^mu2)^k1 -> Y -> iFFT16 over k0,
// inputs carry conj(W1024^lane)^k0.  208 packed instructions where inverse_dec4 has 246.
template <bool PF = (IF_FIR_FFT_TW_PREFETCH != 0)>
__device__ __forceinline__ void inverse_dec4_tan(const cf (&z)[16], cf (&c)[16], const f2v *tb, const f2v *tc, const XAddr &xa, int lane)
{
    cf a[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
        bfly4<true>(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3], a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
    xst16<XROW>(xa.wx, a);
    xld16<8>(xa.rx, a);
    fft16_tw<true, 4, PF>(a, tb + (lane & 3)); // over k1 -> mu1
    xst16<XROW>(xa.wy, a);
    xld16<8>(xa.ry, c);
    fft16_tw<true, 64, PF>(c, tc + lane); // over k0 -> mu0
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

} // namespace if_fir
