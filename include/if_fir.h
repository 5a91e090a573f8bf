/*
 * if_fir.h — C ABI of libif_fir.so: MI355X-native IF-chain FIR filter / decimator for complex IQ streams.
 *
 * Plain C99, no HIP types.  Host code stays C and reaches the CDNA4 kernels through these entry points.
 *
 * WHAT THIS REPLACES IN THE REFERENCE: nothing that exists.  vankxr/qo-100-tools has no filter entry
 * point to bind: `util/if-bandpass-filter/` is an analog LC design
 * (/root/reference/util/if-bandpass-filter/schematic.svg:174-222) and `software/opi-rf-manager/package.json:6-15`
 * lists no DSP dependency.  The surface below is BUILD-DEFINED; only the reference's *conventions* are kept:
 *   - snake_case `module_verb_noun` names and Hungarian argument prefixes
 *     (/root/reference/software/upconverter/src/main.c:572,585  — `ulSamples`, `pfPData`, `fAttenuation`);
 *   - `uint8_t` status, 1 = success, 0 = failure
 *     (/root/reference/software/upconverter/src/f1958.c:12-27, .../src/include/adf4351.h:93-100);
 *   - small enums for modes (/root/reference/software/gpsdo/src/include/ocxo.h:15-19).
 *
 * Semantics: docs/SPEC.md.   y[n] = Σ_k h[k]·x[n-k],  y_D[m] = y[mD],  interleaved float32 I/Q, real float32 taps,
 * T-1 samples of history and the decimation phase carried across calls.
 *
 * Error model: every function that can fail returns 0 and records a message retrievable with
 * if_fir_last_error(); nothing aborts or throws across the boundary; a context stays usable after a failed call.
 * There is NO CPU fallback: without a usable HIP device if_fir_init() fails.
 *
 * Threading: one context = one HIP stream, not re-entrant; distinct contexts may be used from distinct threads.
 */
#ifndef IF_FIR_H
#define IF_FIR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IF_FIR_VERSION_MAJOR 0
#define IF_FIR_VERSION_MINOR 1

#define IF_FIR_MAX_TAPS 4096u
#define IF_FIR_MAX_DECIMATION 64u

typedef struct if_fir_ctx if_fir_ctx_t; /* opaque: device buffers, history, phase, stream, last error */

/* kernel families.  AUTO picks per (T, D). */
enum
{
    IF_FIR_BACKEND_AUTO = 0,
    IF_FIR_BACKEND_HIP_DIRECT = 1,  /* register-blocked sample-stationary direct form, taps in SGPRs            */
    IF_FIR_BACKEND_HIP_TAPSPLIT = 2,/* any T, D: taps staged in LDS, split over 4 lanes, partial sums DPP-reduced */
    IF_FIR_BACKEND_HIP_GENERIC = 3, /* any T, D: one output per thread (simple cross-check kernel)                */
    IF_FIR_BACKEND_HIP_FFT = 4      /* overlap-save, wave-private 4096-point FFT (any T ≤ 4096, any D ≤ 64); AUTO's pick */
};

/* input sample formats (if_fir_set_input_format) */
enum
{
    IF_FIR_INPUT_F32 = 0, /* interleaved float32 I,Q (8 bytes per sample), the default                         */
    IF_FIR_INPUT_I16 = 1  /* interleaved int16 I,Q (4 bytes per sample), value = int16 * 2^-15  */
};

/* window ids for if_bpf_design */
enum
{
    IF_BPF_WINDOW_RECT = 0,
    IF_BPF_WINDOW_HAMMING = 1,
    IF_BPF_WINDOW_HANN = 2,
    IF_BPF_WINDOW_BLACKMAN = 3
};

/* ---- tap generator (host, once; SPEC §4) ------------------------------------------------------ */
/* Windowed-sinc band-pass, odd ulTaps, band [dLow,dHigh] in cycles/sample (0 ≤ dLow < dHigh ≤ 0.5),
 * unity gain at the band centre.  Defaults used by the benchmarks: 0.15–0.25, Blackman. */
uint8_t if_bpf_design(float *pfTaps, uint32_t ulTaps, double dLow, double dHigh, uint32_t ulWindow);

/* Complex taps for channel selection (one-sided band-pass): low-pass prototype of two-sided bandwidth dBandwidth
 * shifted to dCentre (both in cycles/sample, |dCentre| ≤ 0.5); pfTapsIQ receives ulTaps interleaved (re, im) pairs. */
uint8_t if_bpf_design_complex(float *pfTapsIQ, uint32_t ulTaps, double dCentre, double dBandwidth, uint32_t ulWindow);

/* ---- context -------------------------------------------------------------------------------------------------- */
/* Copies the taps to the device, allocates history (and, for if_fir_process, staging buffers sized for
 * ullMaxSamples input samples per call).  lDevice = HIP device ordinal. */
uint8_t if_fir_init(if_fir_ctx_t **ppCtx, const float *pfTaps, uint32_t ulTaps, uint32_t ulDecimation,
                    uint64_t ullMaxSamples, int32_t lDevice);
/* Same with COMPLEX taps (ulTaps interleaved re,im pairs): y[n] = Σ g[k]·x[n-k].  Served by the overlap-save backend at
 * the speed of real taps (its transfer function is complex anyway) and by the generic kernel; the unrolled direct and
 * tap-split kernels take real taps only. */
uint8_t if_fir_init_complex(if_fir_ctx_t **ppCtx, const float *pfTapsIQ, uint32_t ulTaps, uint32_t ulDecimation,
                            uint64_t ullMaxSamples, int32_t lDevice);
void if_fir_destroy(if_fir_ctx_t *pCtx);
/* zero the history and the decimation phase */
uint8_t if_fir_reset(if_fir_ctx_t *pCtx);
uint8_t if_fir_set_backend(if_fir_ctx_t *pCtx, uint32_t ulBackend);
uint32_t if_fir_get_backend(const if_fir_ctx_t *pCtx); /* the resolved (non-AUTO) backend */
/* Input sample format: IF_FIR_INPUT_F32 (default) or IF_FIR_INPUT_I16 (pfIQIn / pDevIn then point at int16 pairs; the
 * conversion is fused into the kernels' loads; overlap-save and generic backends).  Outputs stay float32. */
uint8_t if_fir_set_input_format(if_fir_ctx_t *pCtx, uint32_t ulFormat);
/* NCO fused into the filter (docs/SPEC.md §3.2): the input is mixed with exp(-j*2*pi*f*a), a = absolute
 * sample index since init/reset, f quantised to a 32-bit phase word (if_fir_get_nco returns the value applied).
 * dFreq in cycles/sample, |dFreq| <= 0.5; 0 switches the NCO off.  Takes effect from the next call, as if it had been
 * set since the last reset (the phase is a function of the absolute index).  Overlap-save and generic kernels only. */
uint8_t if_fir_set_nco(if_fir_ctx_t *pCtx, double dFreq);
uint8_t if_fir_get_nco(const if_fir_ctx_t *pCtx, double *pdFreq);
/* expert knob: pick a schedule variant of the direct-form kernels (0 = default, 1..6: workgroup kernel v1 in
 * three tile shapes, the wave kernel with 8- or 16-byte LDS reads and 2 / 4 / 8 tiles per run).  Variants change
 * speed only.  Everything else (diagnostic launches, grid limits, test hooks) lives in the development library, see
 * if_fir_debug.h.
 * Environment: IF_FIR_RCCL_LIBRARY = path of the library providing the nccl* entry points of the multi-channel front
 * (default: librccl.so.1 by name); IF_FIR_MC_TIMEOUT_S = seconds a rank waits for its peers' transfers (default 300). */
uint8_t if_fir_set_tuning(if_fir_ctx_t *pCtx, uint32_t ulVariant);
/* (calls on a stream that is being captured into a hipGraph are refused: the streaming state advances on the host) */
/* run on a caller-owned HIP stream (pass a hipStream_t as void*; NULL = the context's own stream) */
uint8_t if_fir_set_stream(if_fir_ctx_t *pCtx, void *pStream);
/* waits for the context's stream; also fails (0 + message) if the overlap-save kernel's block queue reported an expired
 * bounded wait since the last check -- outputs would then be incomplete; it never should */
uint8_t if_fir_synchronize(if_fir_ctx_t *pCtx);
/* last error message of this context (or of the failed if_fir_init when pCtx is NULL); never NULL */
const char *if_fir_last_error(const if_fir_ctx_t *pCtx);

/* number of output samples a call with ullSamples inputs will produce in the current phase */
uint64_t if_fir_out_count(const if_fir_ctx_t *pCtx, uint64_t ullSamples);

/* ---- filtering ------------------------------------------------------------------------------------------------ */
/* Host pointers: H2D copy + kernel + D2H copy, synchronous.  ullSamples ≤ ullMaxSamples of init. */
uint8_t if_fir_process(if_fir_ctx_t *pCtx, const float *pfIQIn, float *pfIQOut, uint64_t ullSamples,
                       uint64_t *pullOutSamples);
/* Device pointers, 16-byte aligned (the overlap-save backend, AUTO's pick, also takes pointers aligned to one sample:
 * 8 bytes, 4 for int16 input), asynchronous on the context's stream; what the benchmark times.
 * pDevOut must hold if_fir_out_count() samples. */
uint8_t if_fir_process_device(if_fir_ctx_t *pCtx, const void *pDevIn, void *pDevOut, uint64_t ullSamples,
                              uint64_t *pullOutSamples);

/* ---- device-side utilities used by benchmarks and tests ------------------------------------------------------ */
/* Fill pDevIQ with the SPEC §5 synthetic stream of channel ulChannel, samples [ullFirst, ullFirst+ullSamples). */
uint8_t if_fir_synth_device(if_fir_ctx_t *pCtx, void *pDevIQ, uint64_t ullFirst, uint64_t ullSamples,
                            uint32_t ulChannel);
/* Device memory helpers so that a pure-C host needs no HIP headers. */
/* mean(|y|^2) of a device IQ buffer (float32 I,Q), accumulated in float64; synchronous.  The measurement side of a
 * level-control loop; writing the attenuator register stays with the rack controller's daemon. */
uint8_t if_fir_power_device(if_fir_ctx_t *pCtx, const void *pDevIQ, uint64_t ullSamples, double *pdMeanPower);
/* page-locked host memory (hipHostMalloc) for the buffers of if_fir_process: the copies then run at PCIe speed and
 * overlap the kernels; ordinary (pageable) buffers work too, more slowly */
uint8_t if_fir_host_alloc(if_fir_ctx_t *pCtx, void **ppHost, uint64_t ullBytes);
uint8_t if_fir_host_free(if_fir_ctx_t *pCtx, void *pHost);
uint8_t if_fir_dev_alloc(if_fir_ctx_t *pCtx, void **ppDev, uint64_t ullBytes);
uint8_t if_fir_dev_free(if_fir_ctx_t *pCtx, void *pDev);
uint8_t if_fir_dev_upload(if_fir_ctx_t *pCtx, void *pDev, const void *pHost, uint64_t ullBytes);
uint8_t if_fir_dev_download(if_fir_ctx_t *pCtx, void *pHost, const void *pDev, uint64_t ullBytes);
/* "gfx950", CU count, etc.: writes a short description of the context's device */
uint8_t if_fir_device_info(const if_fir_ctx_t *pCtx, char *pszOut, uint32_t ulOutBytes);

/* ---- uniform filter bank (BUILD-DEFINED) -----------------------------------------------------------
 * Channel c = mix-down by pulSlots[c]/16 cycles/sample, the context's prototype taps (real or complex), decimation by the
 * context's decimation: the result of ulChannels contexts with if_fir_set_nco(slot/16.0), computed in ONE pass over the input.
 * The context must have <= 3073 taps, float32 or int16 input, run on the overlap-save backend and have decimation 4 (fs/16
 * channels 4x oversampled; no NCO), 8 (2x oversampled) or 16 (the channel rate: all 16 slots come out of one
 * forward transform; the wanted ones are stored); every other multiple of 4 up to 64 is served through the per-channel forms
 * of if_fir_channelizer_process_device_freq (slot s = centre s/16).  At decimation 8 and 16 the context may carry an NCO: it shifts the WHOLE
 * slot grid (channel c is centred at pulSlots[c]/16 + f_nco: a common fine offset).
 * The context's streaming state (history, decimation phase, sample index) is shared
 * by all channels.  ulChannels 1..16, slots 0..15: any subset, repeats allowed (decimation 16: each slot at most once);
 * ppDevOut[c]: 16-byte aligned device buffers of if_fir_out_count() samples each.  Asynchronous on the context's stream
 * like if_fir_process_device.  (Decimation 8, no NCO on the context: four or more channels on even -- or on odd -- slots, no
 * slot listed twice, are computed together from one pair of 8-point transforms per group, whatever their number; a call is then
 * up to two kernel launches on the context's stream.  Results do not depend on the route beyond float32 rounding.) */
uint8_t if_fir_channelizer_process_device(if_fir_ctx_t *pCtx, uint32_t ulChannels, const uint32_t *pulSlots,
                                          const void *pDevIn, void *const *ppDevOut, uint64_t ullSamples,
                                          uint64_t *pullOutSamples);
/* Channels at ARBITRARY centre frequencies from one pass over the input (round 4; decimation 4, 8, 12, ..., 64 -- any multiple of 4 --,
 * context without NCO, real or complex prototype taps, float32 or int16 input).  Channel c = the prototype centred at pdCentre[c] cycles/sample
 * (|pdCentre[c]| <= 0.5), mixed down and decimated by the context's decimation:
 *     y_c[m] = exp(-j 2 pi f_c a) * sum_k h[k] exp(+j 2 pi g_c k) x[a - k],   a = absolute index of the output's input sample,
 * with g_c = the multiple of 1/4096 nearest to f_c (the overlap-save kernel moves the prototype's response by whole bins of its
 * 4096-point transform) and f_c quantised to a 32-bit phase word like if_fir_set_nco.  For centres on the 1/4096 grid this is
 * exactly what ulChannels contexts with if_fir_set_nco(f_c) compute; off the grid the filter sits at most 1/8192 cycles/sample
 * (0.4 % of a slot) beside the centre the output is mixed to.  Everything else as if_fir_channelizer_process_device
 * (slot s of that call = centre s/16 here). */
uint8_t if_fir_channelizer_process_device_freq(if_fir_ctx_t *pCtx, uint32_t ulChannels, const double *pdCentre,
                                               const void *pDevIn, void *const *ppDevOut, uint64_t ullSamples,
                                               uint64_t *pullOutSamples);

/* ---- multi-channel front (BUILD-DEFINED, the reference has no filter surface) -------------------
 * One process per GPU.  Channel c is filtered by rank if_fir_mc_owner(c, world) = c mod world with its own taps and
 * its own streaming state.  The channel inputs and outputs live on rank 0's GPU; if_fir_mc_process_device() is called
 * by EVERY rank with the same ullSamples and moves them itself in chunks of ~2^24 samples: grouped RCCL sends root ->
 * owners on a transfer stream, the filters of a chunk on a second stream once it has landed, its outputs owners -> root
 * behind the next chunk's scatter; a status word per owner tells the root about a remote filter failure (nobody is left
 * waiting).  With ulWorld == 1 nothing is moved and librccl is never loaded.
 * Bootstrap: rank 0 calls if_fir_mc_unique_id(), the host program hands the 128 bytes to the other ranks by whatever
 * means it has (MPI, a socket, torch.distributed, a file), every rank passes them to if_fir_mc_init().
 * Errors: 0 + if_fir_mc_last_error(); the context stays valid.  Not re-entrant per context. */
typedef struct if_fir_mc_ctx if_fir_mc_ctx_t;
#define IF_FIR_MC_ID_BYTES 128u

uint32_t if_fir_mc_owner(uint32_t ulChannel, uint32_t ulWorld);
uint8_t if_fir_mc_unique_id(uint8_t *pubId /* IF_FIR_MC_ID_BYTES */);
/* pfTaps: ulChannels rows of ulTaps real float32 taps.  ullMaxSamples > 0: per-call limit.  Ranks other than 0 stage two
 * chunks per owned channel (not the whole call).  pubId may be NULL when ulWorld == 1. */
uint8_t if_fir_mc_init(if_fir_mc_ctx_t **ppCtx, uint32_t ulChannels, const float *pfTaps, uint32_t ulTaps,
                       uint32_t ulDecimation, uint64_t ullMaxSamples, int32_t lDevice, uint32_t ulRank,
                       uint32_t ulWorld, const uint8_t *pubId);
void if_fir_mc_destroy(if_fir_mc_ctx_t *pCtx);
uint8_t if_fir_mc_reset(if_fir_mc_ctx_t *pCtx);
/* all channels take IF_FIR_INPUT_F32 or IF_FIR_INPUT_I16 samples (every rank must make the same call) */
uint8_t if_fir_mc_set_input_format(if_fir_mc_ctx_t *pCtx, uint32_t ulFormat);
/* rank 0: ppDevIn[c] / ppDevOut[c] = device pointers on rank 0's GPU for every channel; other ranks may pass NULL.
 * Synchronous: returns when this rank's part (transfers and filters) has finished.  *pullOutSamples: per channel.
 * A failing filter on any rank (or an expired wait of its block queue) makes the call fail on that rank AND on rank 0 (the other ranks complete normally; the
 * transfer protocol is always run to its end, nobody is left waiting); the channel streams are then out of step:
 * call if_fir_mc_reset() on every rank before the next call.  After an RCCL failure the communicator is aborted and the
 * context refuses further calls (destroy it and create a new one); the peers notice through the communicator's
 * asynchronous error state, which their waits poll, or at the latest after IF_FIR_MC_TIMEOUT_S seconds (best effort:
 * the failure paths have run against a stand-in transport only; the protocol itself also over the real librccl in
 * loopback, one process playing all ranks -- no multi-GPU node was available to the build). */
uint8_t if_fir_mc_process_device(if_fir_mc_ctx_t *pCtx, const void *const *ppDevIn, void *const *ppDevOut,
                                 uint64_t ullSamples, uint64_t *pullOutSamples);
/* chunk length of the following calls: 0 = default (about 2^24 samples; with one rank: never split), UINT64_MAX = never split,
 * otherwise a request in samples.  The library rounds the request to the nearest multiple of the context's UNIT = lcm(block
 * advance of the filter's overlap-save kernel -- 3840, 3584, 3072, 2048 or 1024 samples --, twice the decimation) and,
 * where the kernel's block grid follows the decimation phase (every even decimation), makes the first chunk of an off-phase call
 * that many samples longer: the blocks of a chunked call are then the blocks of an unchunked one and the results are
 * bit-identical whatever the chunk and whatever the phase.  Calls are split on the overlap-save backend (AUTO's pick) only: with a
 * channel switched to another backend a call that would be split is refused.  Every rank must make the same call.
 * if_fir_mc_get_chunk_samples reports the chunk in effect (0 = not split) and the unit. */
uint8_t if_fir_mc_set_chunk_samples(if_fir_mc_ctx_t *pCtx, uint64_t ullChunk);
uint8_t if_fir_mc_get_chunk_samples(const if_fir_mc_ctx_t *pCtx, uint64_t *pullChunk, uint64_t *pullUnit);
/* the single-channel context behind a channel this rank owns (NULL otherwise), e.g. for if_fir_set_backend() */
if_fir_ctx_t *if_fir_mc_channel_ctx(if_fir_mc_ctx_t *pCtx, uint32_t ulChannel);
const char *if_fir_mc_last_error(const if_fir_mc_ctx_t *pCtx);

#ifdef __cplusplus
}
#endif
#endif /* IF_FIR_H */
