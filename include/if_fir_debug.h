/*
 * if_fir_debug.h — development hooks of the IF-chain FIR library.  NOT part of the product ABI: these entry points and
 * the tuning variants listed below exist only in libif_fir_dev.so (the same sources built with -DIF_FIR_DEVELOPMENT),
 * which the test-suite and the tools under tools/ load; libif_fir.so exports none of them (tests/test_host.py checks).
 *
 * Development tuning variants of if_fir_set_tuning() (dev library only; the diagnostic ones additionally need
 * IF_FIR_DEBUG=1 in the environment because their results are WRONG by construction):
 *   2000 + k       at most k workgroups for the overlap-save kernel: same results; lets small inputs run through every
 *                  stage of the block queue (tests/test_gpu_parity.py)
 *   1000 + bits    diagnostic launches of the overlap-save kernel (IF_FIR_DEBUG=1): 1 skip the global loads, 2 skip the
 *                  stores, 16 every wave fetches the same block, 32 static block map, 64 one wave per SIMD
 *   1000000 + bits the same with room for more bits (round 3; bits 4, 8, 128 belonged to experiments that are closed and
 *                  removed); 256 no tail phase in short launches; 512 the waves of workgroup 0 count a queue fault and leave as if their bounded wait had
 *                  expired: blocks stay unwritten and if_fir_synchronize must report it (the fault path's test);
 *                  2048 (round 4) the queue's tail phase in launches of up to 16 two-wave rounds; 4096 (round 4) the filter
 *                  bank at decimation 8 without the all-slots form: every channel through the per-channel form (same results
 *                  to tolerance; A/B timing and tests); 8192 (round 4) both slot parities of such a call as two launches instead of one
 *                  16384 (round 5) every wave of a workgroup requests its first block ahead of the table copy (the form up to round 5; since
 *                  then only the first wave of every SIMD does: same results, A/B timing)
 *                  262144 (round 5) no single-round launches: a call of at most one block per wave of the chip fills eight waves per workgroup
 *                  as before instead of one block per wave dealt over all CUs (same results, A/B timing; 131072 is the launcher's own bit)
 *   3000           decimation 2, 6, 10, ..., 62 through the full-rate kernel + selecting store instead of the decimate-by-2 tail (same results to
 *                  tolerance; A/B timing)
 *   4000           the next call fails before anything is launched (IF_FIR_DEBUG=1): lets tests reach callers' error paths
 * Environment (dev library only): IF_FIR_DEBUG=1 IF_FIR_VARIANT=n preselects a variant at if_fir_init.
 * IF_FIR_MC_LOOPBACK=N (dev library only, read by if_fir_mc_init with one rank and >= 2 channels; N = 2..16 virtual ranks, 1 = 2):
 * this process plays all ranks of an N-rank world over a one-rank communicator of the real librccl (every send matched by its
 * receive in the same group, peer = itself): the multi-channel front's whole transfer protocol on a one-GPU box
 * (tools/mc_selfcheck.py).
 */
#ifndef IF_FIR_DEBUG_H
#define IF_FIR_DEBUG_H

#include "if_fir.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Time ulReps back-to-back if_fir_process_device() calls with HIP events on the context's stream, after
 * ulWarmup untimed ones; *pfMsPerCall receives the mean.  History/phase are restored afterwards. */
uint8_t if_fir_time_device(if_fir_ctx_t *pCtx, const void *pDevIn, void *pDevOut, uint64_t ullSamples,
                           uint32_t ulWarmup, uint32_t ulReps, float *pfMsPerCall);
/* first call with pullOut = NULL arms per-wave start/end time stamps for the persistent kernels; later calls copy the
 * last launch's stamps (4 x uint64 per wave) and return the number of words written. */
uint32_t if_fir_debug_stamps(if_fir_ctx_t *pCtx, uint64_t *pullOut, uint32_t ulWords);
/* host-only: the overlap-save kernel's table image (float32, ulOutFloats >= IF_FIR_DEBUG_TABLE_FLOATS) for these taps;
 * returns the number of floats written, 0 if the (taps, decimation) pair is not served by that kernel. */
#define IF_FIR_DEBUG_TABLE_FLOATS 21632u
uint32_t if_fir_debug_fft_tables(const float *pfTaps, uint32_t ulTaps, uint32_t bComplexTaps, uint32_t ulDecimation,
                                 uint32_t ulNcoDelta, float *pfOut, uint32_t ulOutFloats);
/* host-only: a filter bank's table image (ulBank 8 or 16; bank 8: ulParity 0 = the per-channel forms' image, also the all-slots
 * form's for the even slots, 1 = the all-slots form's for the odd slots), IF_FIR_DEBUG_TABLE_FLOATS floats */
uint32_t if_fir_debug_fft_tables_bank(const float *pfTaps, uint32_t ulTaps, uint32_t bComplexTaps, uint32_t ulBank, uint32_t ulParity,
                                      float *pfOut, uint32_t ulOutFloats);
/* host-only: the filter bank's tail for a decimation (4, 8 or 16; bOwnCentres: channels at their own centres, every multiple of 4 up
 * to 64 -- the tail then keeps every (decimation / tail)-th output); 0 = not served */
uint32_t if_fir_debug_bank_tail(uint32_t ulDecimation, uint32_t bOwnCentres);
/* host-only: routing of a decimation-8 filter-bank call on the slot grid: pulOut[0], pulOut[1] = slot masks of the all-slots
 * launches (even / odd slots; 0 = none), pulOut[2] = bit c set: channel c goes through the per-channel form */
uint8_t if_fir_debug_bank_plan(const uint32_t *pulSlots, uint32_t ulChannels, uint32_t *pulOut);
/* host-only: the table image of the odd-decimation kernel (decimation 3, 9, 15, ...: 2 * (3 * 1024 + 2176) = 10496 floats) */
#define IF_FIR_DEBUG_ODD_TABLE_FLOATS 10496u
uint32_t if_fir_debug_fft_tables_odd(const float *pfTaps, uint32_t ulTaps, uint32_t bComplexTaps, uint32_t ulDecimation,
                                     uint32_t ulNcoDelta, float *pfOut, uint32_t ulOutFloats);
/* host-only: block-queue layout of an overlap-save launch: pllOut[6] = blocks per group, groups, static groups per
 * workgroup, 0, upper bound of the global ticket counter, workgroups */
uint8_t if_fir_debug_fft_schedule(uint64_t ullBlocks, uint32_t ulWorkgroups, int64_t *pllOut);
/* bounded waits of the overlap-save kernel's block queue that expired since if_fir_init (a word of the context's
 * queue block; 0 in a healthy run: a wave that gives up leaves its blocks unwritten instead of hanging the device) */
uint8_t if_fir_debug_queue_faults(if_fir_ctx_t *pCtx, uint32_t *pulFaults);
/* host-only: the transfer plan of one rank of the multi-channel front for one call, 8 uint64 per operation {kind 0 send /
 * 1 recv, phase 0 scatter / 1 gather / 2 status, group, peer, channel, chunk, byte offset in rank 0's channel buffer,
 * bytes}; returns the operation count */
uint32_t if_fir_mc_debug_plan(uint32_t ulWorld, uint32_t ulChannels, uint32_t ulRank, uint64_t ullSamples,
                              uint32_t ulInBytes, uint32_t ulTaps, uint32_t ulDecimation, uint64_t ullConsumed,
                              uint64_t ullChunk, uint64_t *pullOut, uint32_t ulMaxOps);

#ifdef __cplusplus
}
#endif
#endif /* IF_FIR_DEBUG_H */
