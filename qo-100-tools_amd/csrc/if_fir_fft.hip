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
// Build: this file is compiled twelve times (csrc/Makefile) -- once per overlap length with -DIF_FIR_FFT_ROWS=4|8|16|32|48 (the
// kernel, its launcher and the explicit instantiation of launch_fft_rows<ROWS>; the 32-row unit also carries the two-partition
// launches), once more per overlap length with -DIF_FIR_FFT_DEC2_UNIT on top (the decimate-by-2 tails' instantiations,
// round 5), once with -DIF_FIR_FFT_ODD (the odd-decimation kernel) and once with none of them (host side:
// tables, routing predicates, launch_fft) -- so that the instantiations compile in parallel.
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

#include "if_fir_fft_dev.h" // complex arithmetic, transforms, lane exchanges, memory helpers, block queue (shared by all units)

namespace if_fir
{

#ifdef IF_FIR_FFT_ODD // ================= odd decimations 3, 9, 15, ..., 63: their own compilation unit =================
#include "if_fir_fft_odd.inc"
#endif // IF_FIR_FFT_ODD

#ifdef IF_FIR_FFT_ROWS // ================= kernel + launchers: the per-overlap-length compilation units =================
#if defined(IF_FIR_FFT_HAZARD_PROBE) && !defined(IF_FIR_FFT_DEC2_UNIT)
#define IF_FIR_FFT_DEC2_UNIT 1 // (the probe instantiates the decimate-by-2 tail)
#endif
// (round 5: the decimate-by-2 tails, CHAN 2 and 3, are instantiated in units of their own, -DIF_FIR_FFT_DEC2_UNIT: more units to compile in
// parallel, and per-family build flags for tools/build_ab.sh; -DIF_FIR_FFT_DEC2_PAIRED, development: those units without the single-read attribute)
#if defined(IF_FIR_FFT_DEC2_UNIT) && defined(IF_FIR_FFT_DEC2_PAIRED)
#define FIR_FFT_KERNEL_ATTR
#else
#define FIR_FFT_KERNEL_ATTR IF_FIR_LDS_SINGLE_READS
#endif
#include "if_fir_fft_kernel.inc"
#undef FIR_FFT_KERNEL_ATTR
template <int ROWS>
hipError_t launch_fft_rows(const LaunchArgs &a); // defined and explicitly instantiated in the unit compiled with IF_FIR_FFT_ROWS = ROWS
hipError_t launch_fft_two_partitions(const LaunchArgs &a); // (in the 32-row unit)
#include "if_fir_fft_launch.inc"
#endif // IF_FIR_FFT_ROWS

#ifdef IF_FIR_FFT_HOST // ================= host side =================
template <int ROWS>
hipError_t launch_fft_rows(const LaunchArgs &a); // defined and explicitly instantiated in the unit compiled with IF_FIR_FFT_ROWS = ROWS
hipError_t launch_fft_two_partitions(const LaunchArgs &a); // (in the 32-row unit)
#include "if_fir_fft_host.inc"
#endif // host side

} // namespace if_fir
