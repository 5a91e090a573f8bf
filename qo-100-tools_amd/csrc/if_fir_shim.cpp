// if_fir_shim.cpp — the thin C-ABI over the HIP kernels (include/if_fir.h).  No CPU fallback: every entry point
// that filters needs a HIP device; failures return 0 with a message in if_fir_last_error().
//
// BUILD-DEFINED boundary: the reference has no filter surface to mirror; conventions (uint8_t
// status 1/0, Hungarian prefixes) follow /root/reference/software/upconverter/src/f1958.c:12-27.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "if_fir.h"
#ifdef IF_FIR_DEVELOPMENT
#include "if_fir_debug.h"
#endif
#include "if_fir_kernels.h"

#define IF_FIR_API extern "C" __attribute__((visibility("default")))

struct if_fir_ctx
{
    int device;
    hipStream_t own_stream;
    hipStream_t stream;
    int T, D;
    int ctaps; // taps are complex (interleaved re,im)
    int in_i16; // input format: 0 = float32 I,Q; 1 = int16 I,Q
    uint32_t backend_req;
    uint32_t backend;
    int variant;
    float *d_taps;
    void *d_hist[2]; // the last hist_len samples of the stream (raw input format), ping-pong
    int hist_len;    // max(T-1, the overlap-save block overlap): see hist_len_for()
    int hist_cur;
    uint64_t consumed;
    uint64_t max_samples;
    void *d_stage_in;
    void *d_stage_out;
    hipStream_t copy_in, copy_out; // if_fir_process on long inputs: H2D / kernel / D2H pipelined over chunks
    hipEvent_t *chunk_ev;          // 2 events per chunk (input landed, outputs ready), created on first use
    uint32_t chunk_ev_count;
    float tone[10];
    float *h_taps; // host copy of the caller's taps (FFT tables are built on demand)
    bool tables_odd;   // d_fft_tables holds the odd-decimation kernel's image (ensure_fft_tables)
    uint32_t nco_word; // SPEC §3.2 phase word (0 = no NCO)
    float *h_eff; // NCO on: effective complex taps g[k] = h[k] e^{+j theta k} (2T floats), else nullptr
    void *d_fft_tables; // overlap-save backend tables (built on first use)
    void *d_fft_tables_bank; // 16-slot filter bank (decimation 16): its own table image (built on first use)
    void *d_queue; // atomic run queue of the persistent kernels
    uint32_t queue_base; // overlap-save launches alternate between two counters: the one the next launch draws from ...
    bool queue_valid;    // ... valid while nobody else (direct kernel, memset) touched the queue words
    void *d_dbg; // diagnostic wave stamps (if_fir_debug_stamps)
    void *d_power; // one float64 accumulator (if_fir_power_device)
    char info[128];
    mutable char err[256];
};

static thread_local char g_init_err[256] = "";

static void set_err(const if_fir_ctx *ctx, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_init_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(ctx, call)                                                                      \
    do                                                                                          \
    {                                                                                           \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
        {                                                                                       \
            set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 0;                                                                           \
        }                                                                                       \
    } while (0)

// taps as the kernels see them: the caller's, or the NCO-shifted complex ones
static inline int eff_ctaps(const if_fir_ctx *ctx) { return ctx->ctaps || ctx->nco_word; }
static inline const float *eff_taps(const if_fir_ctx *ctx) { return ctx->h_eff ? ctx->h_eff : ctx->h_taps; }

// The decimating tail of this context's (taps, decimation): D = F * sub (if_fir::fft_tail; F = 1: none)
static inline int tail_factor(const if_fir_ctx *ctx, int *psub = nullptr)
{
    int F = 1, sub = 1;
    if_fir::fft_tail(ctx->T, ctx->D, &F, &sub);
    if (psub)
        *psub = sub;
    return F;
}

// AUTO: the fastest backend that meets SPEC §3.  Measured over (taps, decimation) from 3 taps to 4095 and decimation
// 1 to 64 (tools/policy_sweep.py, profiles/r01d_policy_sweep.txt, r02_policy_sweep.txt) the overlap-save kernel wins
// wherever it applies -- its cost is that of streaming the data, whatever the tap count -- so it is the pick for every
// filter the library accepts (3074..4096 taps as two partitions).  The unrolled direct form (bit-reproducible order),
// the tap-split kernel (the north_star's wording) and the generic kernel (cross-check) are there on request.
static uint32_t resolve_backend(const if_fir_ctx *ctx, uint32_t req)
{
    if (req != IF_FIR_BACKEND_AUTO)
        return req;
    if (if_fir::fft_supported(ctx->T, ctx->D))
        return IF_FIR_BACKEND_HIP_FFT;
    if (eff_ctaps(ctx) || ctx->in_i16)
        return IF_FIR_BACKEND_HIP_GENERIC;
    return if_fir::direct_supported(ctx->T, ctx->D) ? IF_FIR_BACKEND_HIP_DIRECT : IF_FIR_BACKEND_HIP_TAPSPLIT;
}

// History kept between calls: T-1 samples are what the filter needs; the overlap-save kernel's blocks start a whole
// overlap (4..48 rows of 64 samples) before their first output, and keeping that many makes the first block of a call
// identical to the interior block an unsplit call would have had there (results independent of how a stream is cut
// into calls, as long as the cuts are multiples of the block advance: if_fir_mc_set_chunk_samples).
static int hist_len_for(int T, int D)
{
    const int need = T > 1 ? T - 1 : 0;
    if (!if_fir::fft_supported(T, 1))
        return need;
    if (if_fir::fft_two_partitions(T))
        return 4096; // second partition: 2048 samples of delay + the 2048-sample block overlap
    int ovl = 64 * if_fir::fft_overlap_rows(T, 4); // (the longest overlap any decimation of this filter uses)
    int F = 1, ovlr = 0;
    if (if_fir::fft_odd_tail(T, D, &F, nullptr, &ovlr) && F * 64 * ovlr > ovl)
        ovl = F * 64 * ovlr; // the odd-decimation kernel's blocks (F x 1024 samples) overlap by F x 64 x ovlr input samples
    return ovl > need ? ovl : need;
}

#ifdef IF_FIR_DEVELOPMENT
static bool debug_enabled()
{
    const char *e = getenv("IF_FIR_DEBUG");
    return e && *e && *e != '0';
}
#endif

static bool backend_ok(const if_fir_ctx *ctx, uint32_t b)
{
    if ((eff_ctaps(ctx) || ctx->in_i16) && b != IF_FIR_BACKEND_HIP_FFT && b != IF_FIR_BACKEND_HIP_GENERIC)
        return false; // complex taps / NCO / int16 input: overlap-save and generic kernels only
    switch (b)
    {
    case IF_FIR_BACKEND_HIP_DIRECT:
        return if_fir::direct_supported(ctx->T, ctx->D);
    case IF_FIR_BACKEND_HIP_GENERIC:
    case IF_FIR_BACKEND_HIP_TAPSPLIT:
        return true;
    case IF_FIR_BACKEND_HIP_FFT:
        return if_fir::fft_supported(ctx->T, ctx->D);
    default:
        return false;
    }
}

// twiddles and FFT(taps)/4096 in the kernel's LDS image order, float64 math on the host, once per context
static uint8_t ensure_fft_tables(if_fir_ctx *ctx)
{
    if (ctx->d_fft_tables)
        return 1;
    // 3074..4096 taps run as two partitions (2048 taps + the rest): two table images back to back
    const bool two = if_fir::fft_two_partitions(ctx->T);
    // odd decimations 3, 9, 15, ... (round 4): the image of fir_odd_kernel (development variant 3000 = the full-rate pipeline with
    // a selecting store instead, on the ordinary image; if_fir_set_tuning drops the tables when it crosses that line)
    int oddF = 1;
    const bool odd = if_fir::fft_odd_tail(ctx->T, ctx->D, &oddF, nullptr, nullptr) && ctx->variant != 3000;
    const size_t tab_floats = odd ? (size_t)if_fir::fft_odd_table_floats(oddF) : (size_t)if_fir::FFT_TABLE_FLOATS * (two ? 2 : 1);
    float *tab = (float *)malloc(sizeof(float) * tab_floats);
    if (!tab)
    {
        set_err(ctx, "overlap-save tables: out of host memory");
        return 0;
    }
    ctx->tables_odd = odd;
    // NCO row phasors: per kept output for the decimate-by-4 kernel, per full-rate output for all others
    // int16 input: the kernel leaves the samples unscaled and the table carries the format's 2^-15
    // decimation 4 and its multiples (8, 12, ..., 64: the same tail keeping every sub-th output) take the merged table and
    // NCO row steps of decimation 4; decimation 2 (and 6, 10, ...) and the selecting-store route the plain one
    const int F = tail_factor(ctx);
    const int tabD = F == 4 ? 4 : 1;
    // D = 1 and the selecting store (odd decimations; development variant 3000: decimation 2, 6, 10, ... on it as well): the
    // full-rate pipeline's image
    const int full = (F == 1 || (F == 2 && ctx->variant == 3000)) ? 1 : 0;
    const uint32_t tab_nco = 0u - ctx->nco_word * (F == 4 ? 4u : 1u);
    if (odd)
        if_fir::fft_build_tables_odd(eff_taps(ctx), ctx->T, eff_ctaps(ctx), oddF, 0u - ctx->nco_word * (uint32_t)oddF,
                                     ctx->in_i16 ? 0x1p-15 : 1.0, tab);
    else if (two)
    {
        // two partitions: each a filter of <= 2048 taps with its own table image
        const int step = eff_ctaps(ctx) ? 2 : 1, part = 2048;
        if_fir::fft_build_tables(eff_taps(ctx), part, eff_ctaps(ctx), tabD, tab_nco, ctx->in_i16 ? 0x1p-15 : 1.0, tab, 0, full);
        if_fir::fft_build_tables(eff_taps(ctx) + (size_t)step * part, ctx->T - part, eff_ctaps(ctx), tabD, tab_nco,
                                 ctx->in_i16 ? 0x1p-15 : 1.0, tab + if_fir::FFT_TABLE_FLOATS, 0, full);
    }
    else
        if_fir::fft_build_tables(eff_taps(ctx), ctx->T, eff_ctaps(ctx), tabD, tab_nco, ctx->in_i16 ? 0x1p-15 : 1.0, tab, 0, full);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess)
        e = hipMalloc(&ctx->d_fft_tables, sizeof(float) * tab_floats);
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_fft_tables, tab, sizeof(float) * tab_floats, hipMemcpyHostToDevice);
    free(tab);
    if (e != hipSuccess)
    {
        // never leave an allocated but unfilled image behind: the next call would take it for the tables
        if (ctx->d_fft_tables)
            (void)hipFree(ctx->d_fft_tables);
        ctx->d_fft_tables = nullptr;
        set_err(ctx, "overlap-save tables: upload failed: %s", hipGetErrorString(e));
        return 0;
    }
    return 1;
}

// merged table image of the filter bank at decimation 8 or 16
static uint8_t ensure_bank_tables(if_fir_ctx *ctx)
{
    if (ctx->d_fft_tables_bank)
        return 1;
    // decimation 8: two images back to back -- the bank's own (per-channel forms, and the all-slots form's even slots) and the
    // all-slots form's for the odd slots (round 4)
    const int bank = if_fir::fft_bank_tail(ctx->D, true); // 8 or 16 (decimation 8 / 16 themselves, or the tail behind 24, 32, ..., 64)
    const size_t images = bank == 8 ? 2 : 1;
    float *tab = (float *)malloc(sizeof(float) * if_fir::FFT_TABLE_FLOATS * images);
    if (!tab)
    {
        set_err(ctx, "filter-bank tables: out of host memory");
        return 0;
    }
    // (the NCO's effective complex taps and its per-output phase step, like the single-channel tables)
    if_fir::fft_build_tables(eff_taps(ctx), ctx->T, eff_ctaps(ctx), ctx->D, 0u - ctx->nco_word * (uint32_t)bank,
                             ctx->in_i16 ? 0x1p-15 : 1.0, tab, bank);
    if (images == 2)
        if_fir::fft_build_tables(eff_taps(ctx), ctx->T, eff_ctaps(ctx), ctx->D, 0u - ctx->nco_word * (uint32_t)bank,
                                 ctx->in_i16 ? 0x1p-15 : 1.0, tab + if_fir::FFT_TABLE_FLOATS, bank, 0, 1);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess)
        e = hipMalloc(&ctx->d_fft_tables_bank, sizeof(float) * if_fir::FFT_TABLE_FLOATS * images);
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_fft_tables_bank, tab, sizeof(float) * if_fir::FFT_TABLE_FLOATS * images, hipMemcpyHostToDevice);
    free(tab);
    if (e != hipSuccess)
    {
        if (ctx->d_fft_tables_bank)
            (void)hipFree(ctx->d_fft_tables_bank);
        ctx->d_fft_tables_bank = nullptr;
        set_err(ctx, "filter-bank tables: upload failed: %s", hipGetErrorString(e));
        return 0;
    }
    return 1;
}

static void tone_table(float *t)
{
    for (int i = 0; i < 5; i++)
    {
        const double a1 = 2.0 * M_PI * 0.2 * (double)i, a2 = 2.0 * M_PI * 0.4 * (double)i;
        t[2 * i + 0] = (float)(0.5 * cos(a1) + 0.5 * cos(a2));
        t[2 * i + 1] = (float)(0.5 * sin(a1) + 0.5 * sin(a2));
    }
}

static uint8_t init_common(if_fir_ctx_t **ppCtx, const float *pfTaps, uint32_t ulTaps, uint32_t ulDecimation,
                           uint64_t ullMaxSamples, int32_t lDevice, int ctaps)
{
    if (ppCtx)
        *ppCtx = nullptr;
    if (!ppCtx || !pfTaps)
    {
        set_err(nullptr, "if_fir_init: NULL argument");
        return 0;
    }
    if (ulTaps < 1 || ulTaps > IF_FIR_MAX_TAPS || ulDecimation < 1 || ulDecimation > IF_FIR_MAX_DECIMATION)
    {
        set_err(nullptr, "if_fir_init: taps must be 1..%u and decimation 1..%u (got %u, %u)", IF_FIR_MAX_TAPS,
                IF_FIR_MAX_DECIMATION, ulTaps, ulDecimation);
        return 0;
    }
    const uint32_t tap_floats = ctaps ? 2 * ulTaps : ulTaps;
    for (uint32_t i = 0; i < tap_floats; i++)
        if (!std::isfinite(pfTaps[i]))
        {
            set_err(nullptr, "if_fir_init: tap %u is not finite", ctaps ? i / 2 : i);
            return 0;
        }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
    {
        set_err(nullptr, "if_fir_init: no HIP device available (%s); libif_fir has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return 0;
    }
    if (lDevice < 0 || lDevice >= ndev)
    {
        set_err(nullptr, "if_fir_init: device %d out of range (0..%d)", lDevice, ndev - 1);
        return 0;
    }
    if_fir_ctx *ctx = new (std::nothrow) if_fir_ctx();
    if (!ctx)
    {
        set_err(nullptr, "if_fir_init: out of host memory");
        return 0;
    }
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = lDevice;
    ctx->T = (int)ulTaps;
    ctx->D = (int)ulDecimation;
    ctx->ctaps = ctaps;
    ctx->max_samples = ullMaxSamples;
    tone_table(ctx->tone);
    ctx->h_taps = (float *)malloc(sizeof(float) * tap_floats);
    if (!ctx->h_taps)
    {
        set_err(nullptr, "if_fir_init: out of host memory");
        delete ctx;
        return 0;
    }
    memcpy(ctx->h_taps, pfTaps, sizeof(float) * tap_floats);

#define INIT_TRY(call)                                                                               \
    do                                                                                               \
    {                                                                                                \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
        {                                                                                            \
            set_err(nullptr, "if_fir_init: %s failed: %s", #call, hipGetErrorString(e_));            \
            if_fir_destroy(ctx);                                                                     \
            return 0;                                                                                \
        }                                                                                            \
    } while (0)

    INIT_TRY(hipSetDevice(lDevice));
    hipDeviceProp_t prop;
    INIT_TRY(hipGetDeviceProperties(&prop, lDevice));
    snprintf(ctx->info, sizeof(ctx->info), "%s %s cus=%d clock_mhz=%d lds_per_block=%zu", prop.name, prop.gcnArchName,
             prop.multiProcessorCount, prop.clockRate / 1000, (size_t)prop.sharedMemPerBlock);
    INIT_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    // taps are fetched in whole 16-float blocks by s_load_dwordx16: pad the device copy with zeros to a multiple of 64
    const size_t taps_padded = ((size_t)tap_floats + 63) / 64 * 64;
    INIT_TRY(hipMalloc((void **)&ctx->d_taps, sizeof(float) * taps_padded));
    INIT_TRY(hipMemset(ctx->d_taps, 0, sizeof(float) * taps_padded));
    INIT_TRY(hipMemcpy(ctx->d_taps, pfTaps, sizeof(float) * tap_floats, hipMemcpyHostToDevice));
    INIT_TRY(hipMalloc(&ctx->d_queue, 32));
    INIT_TRY(hipMemset(ctx->d_queue, 0, 32));
    ctx->hist_len = hist_len_for((int)ulTaps, (int)ulDecimation);
    const size_t hist_bytes = 8 * (size_t)(ctx->hist_len > 0 ? ctx->hist_len : 1);
    for (int i = 0; i < 2; i++)
    {
        INIT_TRY(hipMalloc(&ctx->d_hist[i], hist_bytes));
        INIT_TRY(hipMemset(ctx->d_hist[i], 0, hist_bytes));
    }
#undef INIT_TRY
    ctx->backend_req = IF_FIR_BACKEND_AUTO;
    ctx->backend = resolve_backend(ctx, IF_FIR_BACKEND_AUTO);
    if (ctx->backend == IF_FIR_BACKEND_HIP_FFT && !ensure_fft_tables(ctx))
    {
        snprintf(g_init_err, sizeof(g_init_err), "if_fir_init: %s", ctx->err);
        if_fir_destroy(ctx);
        return 0;
    }
#ifdef IF_FIR_DEVELOPMENT
    // development library only: IF_FIR_DEBUG=1 IF_FIR_VARIANT=n preselects a tuning variant
    const char *v = debug_enabled() ? getenv("IF_FIR_VARIANT") : nullptr;
    ctx->variant = v ? atoi(v) : 0;
#endif
    ctx->err[0] = 0;
    *ppCtx = ctx;
    return 1;
}

IF_FIR_API uint8_t if_fir_init(if_fir_ctx_t **ppCtx, const float *pfTaps, uint32_t ulTaps, uint32_t ulDecimation,
                               uint64_t ullMaxSamples, int32_t lDevice)
{
    return init_common(ppCtx, pfTaps, ulTaps, ulDecimation, ullMaxSamples, lDevice, 0);
}

IF_FIR_API uint8_t if_fir_init_complex(if_fir_ctx_t **ppCtx, const float *pfTapsIQ, uint32_t ulTaps,
                                       uint32_t ulDecimation, uint64_t ullMaxSamples, int32_t lDevice)
{
    return init_common(ppCtx, pfTapsIQ, ulTaps, ulDecimation, ullMaxSamples, lDevice, 1);
}

IF_FIR_API void if_fir_destroy(if_fir_ctx_t *pCtx)
{
    if (!pCtx)
        return;
    (void)hipSetDevice(pCtx->device);
    // work queued on a caller's stream still reads the buffers freed below (hipFree waits for the device as well;
    // this keeps the intent explicit and survives a caller that has already destroyed its stream)
    if (pCtx->stream && pCtx->stream != pCtx->own_stream && hipStreamSynchronize(pCtx->stream) != hipSuccess)
        (void)hipGetLastError();
    if (pCtx->own_stream)
    {
        (void)hipStreamSynchronize(pCtx->own_stream);
        (void)hipStreamDestroy(pCtx->own_stream);
    }
    if (pCtx->d_taps)
        (void)hipFree(pCtx->d_taps);
    for (int i = 0; i < 2; i++)
        if (pCtx->d_hist[i])
            (void)hipFree(pCtx->d_hist[i]);
    if (pCtx->copy_in)
        (void)hipStreamDestroy(pCtx->copy_in);
    if (pCtx->copy_out)
        (void)hipStreamDestroy(pCtx->copy_out);
    for (uint32_t i = 0; i < pCtx->chunk_ev_count; i++)
        (void)hipEventDestroy(pCtx->chunk_ev[i]);
    free(pCtx->chunk_ev);
    if (pCtx->d_stage_in)
        (void)hipFree(pCtx->d_stage_in);
    if (pCtx->d_stage_out)
        (void)hipFree(pCtx->d_stage_out);
    if (pCtx->d_dbg)
        (void)hipFree(pCtx->d_dbg);
    if (pCtx->d_power)
        (void)hipFree(pCtx->d_power);
    if (pCtx->d_queue)
        (void)hipFree(pCtx->d_queue);
    if (pCtx->d_fft_tables)
        (void)hipFree(pCtx->d_fft_tables);
    if (pCtx->d_fft_tables_bank)
        (void)hipFree(pCtx->d_fft_tables_bank);
    free(pCtx->h_taps);
    free(pCtx->h_eff);
    delete pCtx;
}

IF_FIR_API const char *if_fir_last_error(const if_fir_ctx_t *pCtx)
{
    return pCtx ? pCtx->err : g_init_err;
}

IF_FIR_API uint8_t if_fir_reset(if_fir_ctx_t *pCtx)
{
    if (!pCtx)
        return 0;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    const size_t hist_bytes = 8 * (size_t)(pCtx->hist_len > 0 ? pCtx->hist_len : 1);
    for (int i = 0; i < 2; i++)
        HIP_TRY(pCtx, hipMemsetAsync(pCtx->d_hist[i], 0, hist_bytes, pCtx->stream));
    pCtx->hist_cur = 0;
    pCtx->consumed = 0;
    return 1;
}

IF_FIR_API uint8_t if_fir_set_backend(if_fir_ctx_t *pCtx, uint32_t ulBackend)
{
    if (!pCtx)
        return 0;
    const uint32_t b = resolve_backend(pCtx, ulBackend);
    if (!backend_ok(pCtx, b))
    {
        set_err(pCtx, "if_fir_set_backend: backend %u does not support taps=%d decimation=%d", ulBackend, pCtx->T,
                pCtx->D);
        return 0;
    }
    if (b == IF_FIR_BACKEND_HIP_FFT && !ensure_fft_tables(pCtx))
        return 0;
    pCtx->backend_req = ulBackend;
    pCtx->backend = b;
    return 1;
}

IF_FIR_API uint32_t if_fir_get_backend(const if_fir_ctx_t *pCtx)
{
    return pCtx ? pCtx->backend : 0;
}

IF_FIR_API uint8_t if_fir_set_tuning(if_fir_ctx_t *pCtx, uint32_t ulVariant)
{
    if (!pCtx)
        return 0;
#ifdef IF_FIR_DEVELOPMENT
    // (if_fir_debug.h) diagnostic launches of the overlap-save kernel skip loads or stores (WRONG results, for timing
    // studies) and 4000 injects a failure: refused unless the process runs with IF_FIR_DEBUG=1
    if (((ulVariant >= 1000 && ulVariant < 2000) || ulVariant == 4000 || ulVariant >= 1000000) && !debug_enabled())
    {
        set_err(pCtx, "if_fir_set_tuning: variant %u is a diagnostic launch (wrong results); set IF_FIR_DEBUG=1 to allow it",
                ulVariant);
        return 0;
    }
#else
    if (ulVariant > 6)
    {
        set_err(pCtx, "if_fir_set_tuning: variant %u is not a schedule variant (0..6); development variants exist in "
                      "libif_fir_dev.so only (if_fir_debug.h)", ulVariant);
        return 0;
    }
#endif
    int tailF = 1;
    (void)if_fir::fft_tail(pCtx->T, pCtx->D, &tailF, nullptr);
    if (((int)ulVariant == 3000) != (pCtx->variant == 3000) && pCtx->d_fft_tables &&
        (if_fir::fft_odd_tail(pCtx->T, pCtx->D, nullptr, nullptr, nullptr) || tailF == 2))
    {
        // the selecting-store route and the odd-decimation kernel take different table images: rebuilt on the next call
        HIP_TRY(pCtx, hipSetDevice(pCtx->device));
        HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
        (void)hipFree(pCtx->d_fft_tables);
        pCtx->d_fft_tables = nullptr;
    }
    pCtx->variant = (int)ulVariant;
    return 1;
}

IF_FIR_API uint8_t if_fir_set_stream(if_fir_ctx_t *pCtx, void *pStream)
{
    if (!pCtx)
        return 0;
    hipStream_t next = pStream ? (hipStream_t)pStream : pCtx->own_stream;
    if (next != pCtx->stream)
    {
        // launches of one context share its run queue, history buffers and tables: they must not overlap, so the work
        // already queued on the old stream is finished before anything goes to the new one
        HIP_TRY(pCtx, hipSetDevice(pCtx->device));
        if (hipStreamSynchronize(pCtx->stream) != hipSuccess) // e.g. the caller has destroyed its old stream already
        {
            (void)hipGetLastError();
            HIP_TRY(pCtx, hipDeviceSynchronize());
        }
    }
    pCtx->stream = next;
    return 1;
}

// The overlap-save kernel's block queue bounds every wait (if_fir_fft_queue.h): a wave that gives up leaves its blocks
// unwritten and counts a fault in the queue block.  That must never happen; if it does, the caller is told here instead of
// being handed incomplete outputs silently.
static uint8_t check_queue_faults(if_fir_ctx *ctx)
{
    // on the context's stream (callers have just synchronized it): a copy on the null stream would also wait for every
    // blocking stream of the process
    uint32_t faults = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&faults, static_cast<const char *>(ctx->d_queue) + 16, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (faults)
    {
        (void)hipMemsetAsync(static_cast<char *>(ctx->d_queue) + 16, 0, 4, ctx->stream);
        ctx->queue_valid = false; // the ticket counters are in an unknown state: the next launch zeroes them itself
        set_err(ctx, "overlap-save block queue: %u bounded wait(s) expired since the last check; outputs of the calls in between "
                     "are incomplete (please report: this is a library defect)", faults);
        return 0;
    }
    return 1;
}

// library-internal (if_fir_mc.cpp): the queue fault counter's device address
extern "C" __attribute__((visibility("hidden"))) const uint32_t *if_fir_internal_fault_word(const if_fir_ctx_t *pCtx)
{
    return pCtx ? reinterpret_cast<const uint32_t *>(static_cast<const char *>(pCtx->d_queue) + 16) : nullptr;
}

IF_FIR_API uint8_t if_fir_synchronize(if_fir_ctx_t *pCtx)
{
    if (!pCtx)
        return 0;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
    return check_queue_faults(pCtx);
}

static inline uint64_t out_count(uint64_t consumed, uint64_t n, uint32_t d, uint32_t *pn0)
{
    const uint64_t n0 = (d - consumed % d) % d;
    if (pn0)
        *pn0 = (uint32_t)n0;
    return (n > n0) ? (n - n0 + d - 1) / d : 0;
}

IF_FIR_API uint64_t if_fir_out_count(const if_fir_ctx_t *pCtx, uint64_t ullSamples)
{
    if (!pCtx)
        return 0;
    return out_count(pCtx->consumed, ullSamples, (uint32_t)pCtx->D, nullptr);
}

// one launch of filter + history kernels; commit = advance the stream state
static uint8_t run_device(if_fir_ctx *ctx, const void *in, void *out, uint64_t n, uint64_t *pout, bool commit,
                          if_fir::ChanArgs *chan = nullptr)
{
    if (n > (uint64_t)1 << 40)
    {
        set_err(ctx, "if_fir_process_device: sample count too large");
        return 0;
    }
#ifdef IF_FIR_DEVELOPMENT
    if (ctx->variant == 4000 && debug_enabled())
    {
        // test hook (dev library, IF_FIR_DEBUG=1): the next call fails before anything is launched -- lets the tests exercise
        // the error paths of callers (the multi-channel front's status word) without breaking a device
        ctx->variant = 0;
        set_err(ctx, "if_fir_process_device: injected failure (tuning variant 4000, test hook)");
        return 0;
    }
#endif
    // the direct-form kernels move 16 bytes per lane; the overlap-save kernel moves one sample per lane and takes any
    // sample-aligned pointer (the chunked multi-channel front hands it pieces that start at odd sample offsets)
    const uintptr_t in_mask = ctx->backend == IF_FIR_BACKEND_HIP_FFT ? (ctx->in_i16 ? 3 : 7) : 15;
    const uintptr_t out_mask = ctx->backend == IF_FIR_BACKEND_HIP_FFT ? 7 : 15;
    if (((uintptr_t)in & in_mask) || ((uintptr_t)out & out_mask))
    {
        set_err(ctx, "if_fir_process_device: device pointers must be %u-byte (input) and %u-byte (output) aligned for this "
                     "backend", (unsigned)in_mask + 1, (unsigned)out_mask + 1);
        return 0;
    }
    uint32_t n0 = 0;
    const uint64_t m = out_count(ctx->consumed, n, (uint32_t)ctx->D, &n0);
    if (pout)
        *pout = m;
    if (n == 0)
        return 1;
    // the streaming state (sample count, decimation phase, history ping-pong, run-queue base) advances on the host with
    // every call and is baked into the launch arguments: a captured graph would replay stale state
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &capture) == hipSuccess && capture != hipStreamCaptureStatusNone)
    {
        set_err(ctx, "if_fir_process_device: the context's stream is being captured into a hipGraph; calls carry host-side "
                     "streaming state and cannot be replayed");
        return 0;
    }
    // (built at init / set_backend / set_nco; a failed rebuild after if_fir_set_nco or if_fir_set_input_format is retried here
    // instead of launching without tables)
    if (ctx->backend == IF_FIR_BACKEND_HIP_FFT && !ensure_fft_tables(ctx))
        return 0;
    if_fir::LaunchArgs a{};
    a.in = in;
    a.out = out;
    a.taps = ctx->d_taps;
    // the kernels other than the overlap-save one look at the T-1 most recent samples: the tail of the buffer
    a.hist_full = ctx->d_hist[ctx->hist_cur];
    a.hist_len = ctx->hist_len;
    a.hist = static_cast<const char *>(a.hist_full) + (size_t)(ctx->hist_len - (ctx->T - 1)) * (ctx->in_i16 ? 4 : 8);
    a.T = ctx->T;
    a.D = ctx->D;
    a.ctaps = eff_ctaps(ctx);
    a.nco_word = ctx->nco_word;
    a.nco_abs0 = (uint32_t)ctx->consumed;
    a.in_i16 = ctx->in_i16;
    a.N = (int64_t)n;
    a.n0 = (int32_t)n0;
    a.M = (int64_t)m;
    a.backend = (int)ctx->backend;
    a.device = ctx->device;
    a.stream = ctx->stream;
    a.fft_tables = ctx->d_fft_tables;
    a.fft_tables_b = (ctx->d_fft_tables && if_fir::fft_two_partitions(ctx->T))
                         ? static_cast<const float *>(ctx->d_fft_tables) + if_fir::FFT_TABLE_FLOATS : nullptr;
    a.queue = ctx->d_queue;
#ifdef IF_FIR_DEVELOPMENT
    a.dbg = ctx->d_dbg;
    // FFT tuning variants (if_fir_debug.h): 1000 + bits / 1000000 + bits = diagnostics, 2000 + k = at most k workgroups
    const bool fft_var = ctx->backend == IF_FIR_BACKEND_HIP_FFT;
    a.diag = (fft_var && ctx->variant >= 1000 && ctx->variant < 2000) ? ctx->variant - 1000
             : (fft_var && ctx->variant >= 1000000 && ctx->variant < 3000000) ? ctx->variant - 1000000 : 0;
    a.grid_limit = (fft_var && ctx->variant > 2000 && ctx->variant < 3000) ? ctx->variant - 2000 : 0;
    a.no_fold = ctx->variant == 3000;
#endif
    if (chan)
    {
        // mix-down phase of every channel at this call's first output: exp(-j 2 pi slot (consumed + n0) / 16)
        // (decimation 16: the arrays are indexed by slot, all 16 are filled)
        const bool slots16 = ctx->D == 16 && !chan->general; // (channels at their own centres: indexed by channel, no slot phases)
        const uint32_t entries = slots16 ? 16u : chan->count;
        for (uint32_t c = 0; c < entries; c++)
        {
            const uint32_t slot = slots16 ? c : chan->slot[c];
            const uint32_t e = (uint32_t)((slot * ((ctx->consumed + n0) & 15u)) & 15u);
            chan->rot0[c][0] = (float)cos(-2.0 * M_PI * (double)e / 16.0);
            chan->rot0[c][1] = (float)sin(-2.0 * M_PI * (double)e / 16.0);
        }
        if (slots16)
        {
            chan->rot_e = (uint32_t)((ctx->consumed + n0) & 15u);
            chan->mask16 = 0;
            for (uint32_t c = 0; c < 16; c++)
                chan->mask16 |= chan->out[c] ? (1u << c) : 0u;
        }
        chan->abs0n0 = (uint32_t)(ctx->consumed + n0);
        a.chan = chan;
        const int fb = if_fir::fft_bank_tail(ctx->D, chan->general != 0);
        if (fb >= 8)
            a.fft_tables = ctx->d_fft_tables_bank;
        if (fb == 8) // (the all-slots form's image for the odd slots)
            a.fft_tables_b = static_cast<const float *>(ctx->d_fft_tables_bank) + if_fir::FFT_TABLE_FLOATS;
    }
    a.queue_base = &ctx->queue_base;
    a.queue_valid = &ctx->queue_valid;
    // the overlap-save kernel updates the history itself (one launch per call) whenever it is launched at all
    const bool fused_history = ctx->backend == IF_FIR_BACKEND_HIP_FFT && m > 0 && ctx->hist_len > 0;
    a.hist_out = fused_history ? ctx->d_hist[ctx->hist_cur ^ 1] : nullptr;
    if (ctx->backend == IF_FIR_BACKEND_HIP_FFT)
        HIP_TRY(ctx, if_fir::launch_fft(a));
    else
        HIP_TRY(ctx, if_fir::launch_fir(a, ctx->variant));
    if (!fused_history)
        HIP_TRY(ctx, if_fir::launch_history(in, ctx->d_hist[ctx->hist_cur], ctx->d_hist[ctx->hist_cur ^ 1], ctx->hist_len,
                                            (int64_t)n, ctx->in_i16, ctx->stream));
    if (commit)
    {
        ctx->hist_cur ^= 1;
        ctx->consumed += n;
    }
    return 1;
}

IF_FIR_API uint8_t if_fir_process_device(if_fir_ctx_t *pCtx, const void *pDevIn, void *pDevOut, uint64_t ullSamples,
                                         uint64_t *pullOutSamples)
{
    if (!pCtx)
        return 0;
    if (pullOutSamples)
        *pullOutSamples = 0;
    if (ullSamples && (!pDevIn || !pDevOut))
    {
        set_err(pCtx, "if_fir_process_device: NULL buffer");
        return 0;
    }
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    return run_device(pCtx, pDevIn, pDevOut, ullSamples, pullOutSamples, true);
}

// Uniform filter bank (BUILD-DEFINED): channel c = the context's real prototype taps applied after a
// mix-down by pulSlots[c]/16 cycles/sample, decimated by 4 -- the same result as ulChannels contexts with
// if_fir_set_nco(slot/16), from ONE pass over the input (one forward transform per block, one small inverse per channel).
// pdFreq != nullptr (if_fir_channelizer_process_device_freq, decimation 4, 8, 12, ..., 64): channel c is centred at pdFreq[c] cycles/sample
// instead of on a slot
static uint8_t channelizer_run(if_fir_ctx_t *pCtx, uint32_t ulChannels, const uint32_t *pulSlots, const double *pdFreq,
                               const void *pDevIn, void *const *ppDevOut, uint64_t ullSamples, uint64_t *pullOutSamples)
{
    if (!pCtx)
        return 0;
    if (pullOutSamples)
        *pullOutSamples = 0;
    if (ulChannels < 1 || ulChannels > (uint32_t)if_fir::CHAN_MAX || (!pulSlots && !pdFreq) || !ppDevOut)
    {
        set_err(pCtx, "if_fir_channelizer_process_device: 1..%d channels with slot (or centre) and output arrays", if_fir::CHAN_MAX);
        return 0;
    }
    // decimation 4, 8, 16: the slot forms; every other multiple of 4 up to 64 runs behind one of their tails keeping every (D / tail)-th
    // output -- through the per-channel general forms, also for channels given as slots (centre bin 256 slot)
    const bool thin = pCtx->D != 4 && pCtx->D != 8 && pCtx->D != 16;
    const int fb = if_fir::fft_bank_tail((int)pCtx->D, pdFreq != nullptr || thin); // the bank's tail: 4, 8 or 16 (0: decimation not served)
    if (pdFreq && (!fb || pCtx->nco_word))
    {
        set_err(pCtx, "if_fir_channelizer_process_device_freq: needs a context with decimation 4, 8, 12, ..., 64 (a multiple of 4) and no "
                      "NCO (every channel carries its own centre frequency)");
        return 0;
    }
    if (!fb || (fb == 4 && pCtx->nco_word) ||
        !if_fir::fft_supported(pCtx->T, pCtx->D) || if_fir::fft_two_partitions(pCtx->T))
    {
        set_err(pCtx, "if_fir_channelizer_process_device: needs <= 3073 taps and a decimation that is a multiple of 4 up to 64 (no NCO "
                      "at 4, 12, 20, ...)");
        return 0;
    }
    if (pCtx->backend != IF_FIR_BACKEND_HIP_FFT)
    {
        set_err(pCtx, "if_fir_channelizer_process_device: the filter bank runs on the overlap-save backend only "
                      "(context is forced to backend %u)", pCtx->backend);
        return 0;
    }
    if_fir::ChanArgs chan{};
    chan.count = ulChannels;
    chan.general = (pdFreq || thin) ? 1u : 0u;
    for (uint32_t c = 0; c < ulChannels; c++)
    {
        if (pdFreq)
        {
            // centre of the filter: the multiple of fs/4096 nearest to the wanted frequency (the overlap-save kernel moves the
            // prototype's response by whole bins of its 4096-point transform); mix-down: the wanted frequency itself, as a 32-bit
            // phase word
            if (!(pdFreq[c] >= -0.5 && pdFreq[c] <= 0.5) || (ullSamples && !ppDevOut[c]) || ((uintptr_t)ppDevOut[c] & 15))
            {
                set_err(pCtx, "if_fir_channelizer_process_device_freq: channel %u: centre must be within +-0.5 cycles/sample and "
                              "the output a 16-byte aligned device pointer", c);
                return 0;
            }
            chan.bin[c] = (uint32_t)(llround(pdFreq[c] * 4096.0) & 4095);
            chan.pword[c] = (uint32_t)(int64_t)llround(pdFreq[c] * 4294967296.0);
            chan.out[c] = (float2 *)ppDevOut[c];
            for (int m0 = 1; m0 < 16; m0++) // W4096^(m0 bin), the wave-uniform factor beside the gathered table entry (decimation 4: m0 < 4, 8: < 8)
            {
                const double a = -2.0 * M_PI * (double)((m0 * chan.bin[c]) & 4095u) / 4096.0;
                chan.tw[c][2 * (m0 - 1) + 0] = (float)cos(a);
                chan.tw[c][2 * (m0 - 1) + 1] = (float)sin(a);
            }
            continue;
        }
        if (pulSlots[c] > 15 || (ullSamples && !ppDevOut[c]) || ((uintptr_t)ppDevOut[c] & 15))
        {
            set_err(pCtx, "if_fir_channelizer_process_device: channel %u: slot must be 0..15 and the output a 16-byte "
                          "aligned device pointer", c);
            return 0;
        }
        if (pCtx->D == 16)
        {
            // channel rate: the kernel computes all 16 slots from one forward transform and stores the wanted ones;
            // its arrays are indexed by slot
            if (chan.out[pulSlots[c]])
            {
                set_err(pCtx, "if_fir_channelizer_process_device: slot %u is listed twice (decimation 16 takes each slot once)",
                        pulSlots[c]);
                return 0;
            }
            chan.out[pulSlots[c]] = (float2 *)ppDevOut[c];
            continue;
        }
        chan.slot[c] = pulSlots[c];
        chan.out[c] = (float2 *)ppDevOut[c];
        // decimation 8: a slot is the centre bin 256 slot; the context's NCO (a common fine offset: the tables are built from
        // its effective complex taps) joins the slot's mix-down word
        chan.bin[c] = 256u * pulSlots[c];
        chan.pword[c] = (pulSlots[c] << 28) + pCtx->nco_word;
        for (int m0 = 1; m0 < 16; m0++) // decimation 4 uses the first three, 8 the first seven
        {
            const double a = -2.0 * M_PI * (double)((m0 * pulSlots[c]) & 15u) / 16.0; // W16^(m0 slot) = W4096^(m0 bin)
            chan.tw[c][2 * (m0 - 1) + 0] = (float)cos(a);
            chan.tw[c][2 * (m0 - 1) + 1] = (float)sin(a);
        }
    }
    if (ullSamples && !pDevIn)
    {
        set_err(pCtx, "if_fir_channelizer_process_device: NULL input");
        return 0;
    }
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    if (!ensure_fft_tables(pCtx) || (fb != 4 && !ensure_bank_tables(pCtx)))
        return 0;
    return run_device(pCtx, pDevIn, ppDevOut[0], ullSamples, pullOutSamples, true, &chan);
}

IF_FIR_API uint8_t if_fir_channelizer_process_device(if_fir_ctx_t *pCtx, uint32_t ulChannels, const uint32_t *pulSlots,
                                                     const void *pDevIn, void *const *ppDevOut, uint64_t ullSamples,
                                                     uint64_t *pullOutSamples)
{
    return channelizer_run(pCtx, ulChannels, pulSlots, nullptr, pDevIn, ppDevOut, ullSamples, pullOutSamples);
}

// Channels at arbitrary centre frequencies from ONE pass over the input (round 4; decimation 8): see include/if_fir.h
IF_FIR_API uint8_t if_fir_channelizer_process_device_freq(if_fir_ctx_t *pCtx, uint32_t ulChannels, const double *pdCentre,
                                                          const void *pDevIn, void *const *ppDevOut, uint64_t ullSamples,
                                                          uint64_t *pullOutSamples)
{
    if (pCtx && !pdCentre)
    {
        set_err(pCtx, "if_fir_channelizer_process_device_freq: NULL centre array");
        if (pullOutSamples)
            *pullOutSamples = 0;
        return 0;
    }
    return channelizer_run(pCtx, ulChannels, nullptr, pdCentre, pDevIn, ppDevOut, ullSamples, pullOutSamples);
}

IF_FIR_API uint8_t if_fir_process(if_fir_ctx_t *pCtx, const float *pfIQIn, float *pfIQOut, uint64_t ullSamples,
                                  uint64_t *pullOutSamples)
{
    if (!pCtx)
        return 0;
    if (pullOutSamples)
        *pullOutSamples = 0;
    if (ullSamples == 0)
        return 1;
    if (!pfIQIn || !pfIQOut)
    {
        set_err(pCtx, "if_fir_process: NULL buffer");
        return 0;
    }
    if (ullSamples > pCtx->max_samples)
    {
        set_err(pCtx, "if_fir_process: %llu samples exceed the ullMaxSamples=%llu given to if_fir_init",
                (unsigned long long)ullSamples, (unsigned long long)pCtx->max_samples);
        return 0;
    }
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    if (!pCtx->d_stage_in || !pCtx->d_stage_out)
    {
        // both or neither: a context left with one buffer would hand a null pointer to the kernels on the next call
        void *si = nullptr, *so = nullptr;
        HIP_TRY(pCtx, hipMalloc(&si, 8 * pCtx->max_samples));
        const hipError_t e2 = hipMalloc(&so, 8 * (pCtx->max_samples / pCtx->D + 1));
        if (e2 != hipSuccess)
        {
            (void)hipFree(si);
            HIP_TRY(pCtx, e2);
        }
        pCtx->d_stage_in = si;
        pCtx->d_stage_out = so;
    }
    const uint64_t isz = pCtx->in_i16 ? 4 : 8;
    // Long inputs: chunks of ~2^22 samples flow through three streams (copy in, kernels, copy out), so the transfer of
    // chunk i+1 and the return of chunk i-1 overlap the filtering of chunk i (PCIe is full duplex; with pinned host
    // buffers, if_fir_host_alloc, the copies run at link speed).  A chunk is a multiple of 4 D samples: its device
    // pointers stay 16-byte aligned and it produces exactly chunk / D outputs whatever the decimation phase.
    const uint64_t unit = 4ull * (uint64_t)pCtx->D;
    uint64_t chunk = (((uint64_t)1 << 22) / unit) * unit;
    if (chunk < unit)
        chunk = unit;
    while (ullSamples / chunk > 255)
        chunk *= 2;
    if (ullSamples < 2 * chunk)
    {
        HIP_TRY(pCtx, hipMemcpyAsync(pCtx->d_stage_in, pfIQIn, isz * ullSamples, hipMemcpyHostToDevice, pCtx->stream));
        uint64_t m = 0;
        if (!run_device(pCtx, pCtx->d_stage_in, pCtx->d_stage_out, ullSamples, &m, true))
            return 0;
        if (m)
            HIP_TRY(pCtx, hipMemcpyAsync(pfIQOut, pCtx->d_stage_out, 8 * m, hipMemcpyDeviceToHost, pCtx->stream));
        HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
        if (!check_queue_faults(pCtx))
            return 0;
        if (pullOutSamples)
            *pullOutSamples = m;
        return 1;
    }
    const uint32_t nchunks = (uint32_t)((ullSamples + chunk - 1) / chunk);
    if (!pCtx->copy_in)
    {
        HIP_TRY(pCtx, hipStreamCreateWithFlags(&pCtx->copy_in, hipStreamNonBlocking));
        HIP_TRY(pCtx, hipStreamCreateWithFlags(&pCtx->copy_out, hipStreamNonBlocking));
    }
    if (pCtx->chunk_ev_count < 2 * nchunks)
    {
        hipEvent_t *ev = (hipEvent_t *)realloc(pCtx->chunk_ev, sizeof(hipEvent_t) * 2 * nchunks);
        if (!ev)
        {
            set_err(pCtx, "if_fir_process: out of host memory");
            return 0;
        }
        pCtx->chunk_ev = ev;
        while (pCtx->chunk_ev_count < 2 * nchunks)
        {
            HIP_TRY(pCtx, hipEventCreateWithFlags(&pCtx->chunk_ev[pCtx->chunk_ev_count], hipEventDisableTiming));
            pCtx->chunk_ev_count++;
        }
    }
    // the copy streams must not run ahead of work already queued on the context's stream (earlier calls)
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
    uint64_t done_in = 0, done_out = 0;
    const auto pipeline = [&]() -> uint8_t {
    for (uint32_t c = 0; c < nchunks; c++)
    {
        const uint64_t len = (ullSamples - done_in) < chunk ? (ullSamples - done_in) : chunk;
        char *d_in = (char *)pCtx->d_stage_in + isz * done_in;
        char *d_out = (char *)pCtx->d_stage_out + 8 * done_out;
        HIP_TRY(pCtx, hipMemcpyAsync(d_in, (const char *)pfIQIn + isz * done_in, isz * len, hipMemcpyHostToDevice,
                                     pCtx->copy_in));
        HIP_TRY(pCtx, hipEventRecord(pCtx->chunk_ev[2 * c], pCtx->copy_in));
        HIP_TRY(pCtx, hipStreamWaitEvent(pCtx->stream, pCtx->chunk_ev[2 * c], 0));
        uint64_t m = 0;
        if (!run_device(pCtx, d_in, d_out, len, &m, true))
            return 0;
        HIP_TRY(pCtx, hipEventRecord(pCtx->chunk_ev[2 * c + 1], pCtx->stream));
        HIP_TRY(pCtx, hipStreamWaitEvent(pCtx->copy_out, pCtx->chunk_ev[2 * c + 1], 0));
        if (m)
            HIP_TRY(pCtx, hipMemcpyAsync(pfIQOut + 2 * done_out, d_out, 8 * m, hipMemcpyDeviceToHost, pCtx->copy_out));
        done_in += len;
        done_out += m;
    }
    return 1;
    };
    if (!pipeline())
    {
        // a chunk failed: nothing of this call may still be reading or writing the caller's buffers when it returns
        // (the error text of the failing step stays in place)
        (void)hipStreamSynchronize(pCtx->copy_in);
        (void)hipStreamSynchronize(pCtx->stream);
        (void)hipStreamSynchronize(pCtx->copy_out);
        return 0;
    }
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->copy_out));
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
    if (!check_queue_faults(pCtx))
        return 0;
    if (pullOutSamples)
        *pullOutSamples = done_out;
    return 1;
}

#ifdef IF_FIR_DEVELOPMENT
// Host-only: the overlap-save kernel's LDS table image (twiddles, H or the merged G table, NCO row phasors) for a set
// of taps, as fft_build_tables() computes it in float64.  Needs no device: the CPU tests check it against numpy.
IF_FIR_API uint32_t if_fir_debug_fft_tables(const float *pfTaps, uint32_t ulTaps, uint32_t bComplexTaps,
                                            uint32_t ulDecimation, uint32_t ulNcoDelta, float *pfOut, uint32_t ulOutFloats)
{
    if (!pfTaps || !pfOut || ulOutFloats < (uint32_t)if_fir::FFT_TABLE_FLOATS ||
        !if_fir::fft_supported((int)ulTaps, (int)ulDecimation) || if_fir::fft_two_partitions((int)ulTaps))
        return 0; // (a two-partition filter is two such images, one per partition of <= 2048 taps)
    // (the image the library would upload for this pair: decimation 4 and its multiples take the merged table, D = 1 and the
    // selecting store the full-rate pipeline's, decimation 2, 6, 10, ... the plain one)
    int F = 1;
    if_fir::fft_tail((int)ulTaps, (int)ulDecimation, &F, nullptr);
    if_fir::fft_build_tables(pfTaps, (int)ulTaps, bComplexTaps ? 1 : 0, F == 4 ? 4 : 1, ulNcoDelta, 1.0, pfOut, 0, F == 1 ? 1 : 0);
    return (uint32_t)if_fir::FFT_TABLE_FLOATS;
}

// Host-only: the filter bank's table images (bank = 8: ulParity 0 the per-channel forms' and the all-slots form's even slots, 1 the
// all-slots form's odd slots; bank = 16: the 16-slot image) and the routing of a decimation-8 call
IF_FIR_API uint32_t if_fir_debug_fft_tables_bank(const float *pfTaps, uint32_t ulTaps, uint32_t bComplexTaps, uint32_t ulBank,
                                                 uint32_t ulParity, float *pfOut, uint32_t ulOutFloats)
{
    if (!pfTaps || !pfOut || ulOutFloats < (uint32_t)if_fir::FFT_TABLE_FLOATS || (ulBank != 8 && ulBank != 16) ||
        !if_fir::fft_supported((int)ulTaps, (int)ulBank) || if_fir::fft_two_partitions((int)ulTaps))
        return 0;
    if_fir::fft_build_tables(pfTaps, (int)ulTaps, bComplexTaps ? 1 : 0, (int)ulBank, 0u, 1.0, pfOut, (int)ulBank, 0, ulParity ? 1 : 0);
    return (uint32_t)if_fir::FFT_TABLE_FLOATS;
}
IF_FIR_API uint32_t if_fir_debug_bank_tail(uint32_t ulDecimation, uint32_t bOwnCentres)
{
    return (uint32_t)if_fir::fft_bank_tail((int)ulDecimation, bOwnCentres != 0);
}
IF_FIR_API uint8_t if_fir_debug_bank_plan(const uint32_t *pulSlots, uint32_t ulChannels, uint32_t *pulOut)
{
    if (!pulSlots || !pulOut || ulChannels < 1 || ulChannels > (uint32_t)if_fir::CHAN_MAX)
        return 0;
    if_fir::fft_bank8_plan(pulSlots, ulChannels, true, pulOut, pulOut + 2);
    return 1;
}

// Host-only: the table image of the odd-decimation kernel (fir_odd_kernel, F = 3) for a set of taps
IF_FIR_API uint32_t if_fir_debug_fft_tables_odd(const float *pfTaps, uint32_t ulTaps, uint32_t bComplexTaps, uint32_t ulDecimation,
                                                uint32_t ulNcoDelta, float *pfOut, uint32_t ulOutFloats)
{
    int F = 1;
    if (!pfTaps || !pfOut || !if_fir::fft_odd_tail((int)ulTaps, (int)ulDecimation, &F, nullptr, nullptr) ||
        ulOutFloats < (uint32_t)if_fir::fft_odd_table_floats(F))
        return 0;
    if_fir::fft_build_tables_odd(pfTaps, (int)ulTaps, bComplexTaps ? 1 : 0, F, ulNcoDelta, 1.0, pfOut);
    return (uint32_t)if_fir::fft_odd_table_floats(F);
}

// bounded waits of the block queue that expired (if_fir_fft_queue.h): word 4 of the queue block
IF_FIR_API uint8_t if_fir_debug_queue_faults(if_fir_ctx_t *pCtx, uint32_t *pulFaults)
{
    if (!pCtx || !pulFaults)
        return 0;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
    HIP_TRY(pCtx, hipMemcpy(pulFaults, static_cast<const char *>(pCtx->d_queue) + 16, 4, hipMemcpyDeviceToHost));
    return 1;
}

// Host-only: the block-queue layout the overlap-save launcher would use for nblocks blocks on at most ulWorkgroups
// workgroups; pllOut receives blocks per group, groups, static groups per workgroup, 0, ticket bound, workgroups.  The
// CPU tests replay the queue under random interleavings and check that every block is handed out exactly once.
IF_FIR_API uint8_t if_fir_debug_fft_schedule(uint64_t ullBlocks, uint32_t ulWorkgroups, int64_t *pllOut)
{
    if (!pllOut || !ullBlocks || !ulWorkgroups)
        return 0;
    if_fir::FftSchedule s;
    if_fir::fft_schedule((int64_t)ullBlocks, (int64_t)ulWorkgroups, s);
    pllOut[0] = s.RA;
    pllOut[1] = s.nA;
    pllOut[2] = s.RB;
    pllOut[3] = s.nB;
    pllOut[4] = s.tickets;
    pllOut[5] = s.wgs;
    return 1;
}

#endif // IF_FIR_DEVELOPMENT

// Mean power of a device IQ buffer, mean(|y|^2): what a control loop feeds back into an attenuator (the
// reference's rack controller sets the IF attenuation through I2C register 0x20, lib/upconverter.js:176-187; the
// register write stays with the daemon).  Synchronous (returns the number).
IF_FIR_API uint8_t if_fir_power_device(if_fir_ctx_t *pCtx, const void *pDevIQ, uint64_t ullSamples, double *pdMeanPower)
{
    if (!pCtx || !pdMeanPower)
        return 0;
    *pdMeanPower = 0.0;
    if (ullSamples == 0)
        return 1;
    if (!pDevIQ || ((uintptr_t)pDevIQ & 15))
    {
        set_err(pCtx, "if_fir_power_device: the buffer must be a 16-byte aligned device pointer");
        return 0;
    }
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    if (!pCtx->d_power)
        HIP_TRY(pCtx, hipMalloc(&pCtx->d_power, sizeof(double)));
    HIP_TRY(pCtx, if_fir::launch_power(pDevIQ, ullSamples, (double *)pCtx->d_power, pCtx->stream));
    double sum = 0.0;
    HIP_TRY(pCtx, hipMemcpyAsync(&sum, pCtx->d_power, sizeof(double), hipMemcpyDeviceToHost, pCtx->stream));
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
    *pdMeanPower = sum / (double)ullSamples;
    return 1;
}

// pinned host memory for if_fir_process without HIP headers on the caller's side
IF_FIR_API uint8_t if_fir_host_alloc(if_fir_ctx_t *pCtx, void **ppHost, uint64_t ullBytes)
{
    if (!pCtx || !ppHost)
        return 0;
    *ppHost = nullptr;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipHostMalloc(ppHost, ullBytes ? ullBytes : 16, hipHostMallocDefault));
    return 1;
}

IF_FIR_API uint8_t if_fir_host_free(if_fir_ctx_t *pCtx, void *pHost)
{
    if (!pCtx)
        return 0;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipHostFree(pHost));
    return 1;
}

IF_FIR_API uint8_t if_fir_synth_device(if_fir_ctx_t *pCtx, void *pDevIQ, uint64_t ullFirst, uint64_t ullSamples,
                                       uint32_t ulChannel)
{
    if (!pCtx)
        return 0;
    if (ullSamples && !pDevIQ)
    {
        set_err(pCtx, "if_fir_synth_device: NULL buffer");
        return 0;
    }
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, if_fir::launch_synth(pDevIQ, ullFirst, ullSamples, ulChannel, pCtx->tone, pCtx->stream));
    return 1;
}

#ifdef IF_FIR_DEVELOPMENT
IF_FIR_API uint8_t if_fir_time_device(if_fir_ctx_t *pCtx, const void *pDevIn, void *pDevOut, uint64_t ullSamples,
                                      uint32_t ulWarmup, uint32_t ulReps, float *pfMsPerCall)
{
    if (!pCtx)
        return 0;
    if (!pDevIn || !pDevOut || !pfMsPerCall || ulReps == 0 || ullSamples == 0)
    {
        set_err(pCtx, "if_fir_time_device: bad argument");
        return 0;
    }
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    hipEvent_t ev0, ev1;
    HIP_TRY(pCtx, hipEventCreate(&ev0));
    HIP_TRY(pCtx, hipEventCreate(&ev1));
    uint8_t ok = 1;
    for (uint32_t i = 0; i < ulWarmup && ok; i++)
        ok = run_device(pCtx, pDevIn, pDevOut, ullSamples, nullptr, false);
    if (ok && hipEventRecord(ev0, pCtx->stream) != hipSuccess)
        ok = 0;
    for (uint32_t i = 0; i < ulReps && ok; i++)
        ok = run_device(pCtx, pDevIn, pDevOut, ullSamples, nullptr, false);
    if (ok && hipEventRecord(ev1, pCtx->stream) != hipSuccess)
        ok = 0;
    if (ok && hipEventSynchronize(ev1) != hipSuccess)
        ok = 0;
    float ms = 0.f;
    if (ok && hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess)
        ok = 0;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    if (!ok)
    {
        if (!pCtx->err[0])
            set_err(pCtx, "if_fir_time_device: HIP event timing failed");
        return 0;
    }
    *pfMsPerCall = ms / (float)ulReps;
    return 1;
}

#endif // IF_FIR_DEVELOPMENT

IF_FIR_API uint8_t if_fir_dev_alloc(if_fir_ctx_t *pCtx, void **ppDev, uint64_t ullBytes)
{
    if (!pCtx || !ppDev)
        return 0;
    *ppDev = nullptr;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipMalloc(ppDev, ullBytes ? ullBytes : 16));
    return 1;
}

IF_FIR_API uint8_t if_fir_dev_free(if_fir_ctx_t *pCtx, void *pDev)
{
    if (!pCtx)
        return 0;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipFree(pDev));
    return 1;
}

IF_FIR_API uint8_t if_fir_dev_upload(if_fir_ctx_t *pCtx, void *pDev, const void *pHost, uint64_t ullBytes)
{
    if (!pCtx)
        return 0;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipMemcpyAsync(pDev, pHost, ullBytes, hipMemcpyHostToDevice, pCtx->stream));
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
    return 1;
}

IF_FIR_API uint8_t if_fir_dev_download(if_fir_ctx_t *pCtx, void *pHost, const void *pDev, uint64_t ullBytes)
{
    if (!pCtx)
        return 0;
    HIP_TRY(pCtx, hipSetDevice(pCtx->device));
    HIP_TRY(pCtx, hipMemcpyAsync(pHost, pDev, ullBytes, hipMemcpyDeviceToHost, pCtx->stream));
    HIP_TRY(pCtx, hipStreamSynchronize(pCtx->stream));
    return 1;
}

IF_FIR_API uint8_t if_fir_device_info(const if_fir_ctx_t *pCtx, char *pszOut, uint32_t ulOutBytes)
{
    if (!pCtx || !pszOut || !ulOutBytes)
        return 0;
    snprintf(pszOut, ulOutBytes, "%s", pCtx->info);
    return 1;
}

#ifdef IF_FIR_DEVELOPMENT
// Diagnostics: the first call (pullOut == NULL or ulWords == 0) arms per-wave start/end stamps for the persistent
// direct kernel; later calls copy the stamps of the last launch (4 x uint64 per wave: realtime start/end in 10 ns
// ticks, shader clock start/end) and return the number of uint64 words written.
IF_FIR_API uint32_t if_fir_debug_stamps(if_fir_ctx_t *pCtx, uint64_t *pullOut, uint32_t ulWords)
{
    if (!pCtx)
        return 0;
    const uint32_t max_words = 8192 * 4;
    if (hipSetDevice(pCtx->device) != hipSuccess)
        return 0;
    if (!pCtx->d_dbg)
    {
        if (hipMalloc(&pCtx->d_dbg, sizeof(uint64_t) * max_words) != hipSuccess)
        {
            pCtx->d_dbg = nullptr;
            return 0;
        }
        (void)hipMemset(pCtx->d_dbg, 0, sizeof(uint64_t) * max_words);
    }
    if (!pullOut || !ulWords)
        return 0;
    const uint32_t n = ulWords < max_words ? ulWords : max_words;
    if (hipStreamSynchronize(pCtx->stream) != hipSuccess)
        return 0;
    if (hipMemcpy(pullOut, pCtx->d_dbg, sizeof(uint64_t) * n, hipMemcpyDeviceToHost) != hipSuccess)
        return 0;
    return n;
}

#endif // IF_FIR_DEVELOPMENT

// Input sample format: IF_FIR_INPUT_F32 (default) or IF_FIR_INPUT_I16 = interleaved int16 I,Q with
// value = int16 * 2^-15, converted inside the kernels' loads (overlap-save and generic backends).  Must be chosen
// before the first sample is processed (or right after if_fir_reset): the history buffer holds samples in the
// input format.
// SPEC §3.2.  The kernels see complex taps g[k] = h[k] e^{+j theta k} and rotate their outputs; nothing else changes.
IF_FIR_API uint8_t if_fir_set_nco(if_fir_ctx_t *pCtx, double dFreq)
{
    if (!pCtx)
        return 0;
    if (!std::isfinite(dFreq) || std::fabs(dFreq) > 0.5)
    {
        set_err(pCtx, "if_fir_set_nco: frequency must be within +-0.5 cycles/sample (got %g)", dFreq);
        return 0;
    }
    const uint32_t word = (uint32_t)(int64_t)std::llround(dFreq * 4294967296.0); // mod 2^32 (two's complement)
    if (word == pCtx->nco_word)
        return 1;
    const uint32_t T = (uint32_t)pCtx->T;
    float *eff = nullptr;
    if (word)
    {
        eff = (float *)malloc(sizeof(float) * 2 * T);
        if (!eff)
        {
            set_err(pCtx, "if_fir_set_nco: out of host memory");
            return 0;
        }
        for (uint32_t k = 0; k < T; k++)
        {
            const double a = 6.283185307179586476925286766559 * ((double)(uint32_t)(word * k) / 4294967296.0);
            const double hr = pCtx->ctaps ? (double)pCtx->h_taps[2 * k] : (double)pCtx->h_taps[k];
            const double hi = pCtx->ctaps ? (double)pCtx->h_taps[2 * k + 1] : 0.0;
            eff[2 * k + 0] = (float)(hr * cos(a) - hi * sin(a));
            eff[2 * k + 1] = (float)(hr * sin(a) + hi * cos(a));
        }
    }
    // would the requested backend still apply?  (decide before anything is changed)
    const uint32_t old_word = pCtx->nco_word;
    float *old_eff = pCtx->h_eff;
    pCtx->nco_word = word;
    pCtx->h_eff = eff;
    const uint32_t b = resolve_backend(pCtx, pCtx->backend_req);
    if (!backend_ok(pCtx, b))
    {
        pCtx->nco_word = old_word;
        pCtx->h_eff = old_eff;
        free(eff);
        set_err(pCtx, "if_fir_set_nco: backend %u has real-tap arithmetic only (use AUTO, HIP_FFT or HIP_GENERIC)",
                pCtx->backend_req);
        return 0;
    }
    // kernels in flight still read the old taps and tables
    hipError_t e = hipSetDevice(pCtx->device);
    if (e == hipSuccess)
        e = hipStreamSynchronize(pCtx->stream);
    float *d_new = nullptr;
    const size_t floats = eff_ctaps(pCtx) ? 2 * (size_t)T : (size_t)T, padded = (floats + 63) / 64 * 64;
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_new, sizeof(float) * padded);
    if (e == hipSuccess)
        e = hipMemset(d_new, 0, sizeof(float) * padded);
    if (e == hipSuccess)
        e = hipMemcpy(d_new, eff_taps(pCtx), sizeof(float) * floats, hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        if (d_new)
            (void)hipFree(d_new);
        pCtx->nco_word = old_word;
        pCtx->h_eff = old_eff;
        free(eff);
        set_err(pCtx, "if_fir_set_nco: %s", hipGetErrorString(e));
        return 0;
    }
    (void)hipFree(pCtx->d_taps);
    pCtx->d_taps = d_new;
    free(old_eff);
    if (pCtx->d_fft_tables) // H and the row phasors depend on the word: rebuilt on demand
    {
        (void)hipFree(pCtx->d_fft_tables);
        pCtx->d_fft_tables = nullptr;
    }
    if (pCtx->d_fft_tables_bank)
    {
        (void)hipFree(pCtx->d_fft_tables_bank);
        pCtx->d_fft_tables_bank = nullptr;
    }
    pCtx->backend = b;
    if (b == IF_FIR_BACKEND_HIP_FFT && !ensure_fft_tables(pCtx))
        return 0;
    return 1;
}

IF_FIR_API uint8_t if_fir_get_nco(const if_fir_ctx_t *pCtx, double *pdFreq)
{
    if (!pCtx || !pdFreq)
        return 0;
    *pdFreq = (double)(int32_t)pCtx->nco_word / 4294967296.0; // the quantised frequency actually applied
    return 1;
}

IF_FIR_API uint8_t if_fir_set_input_format(if_fir_ctx_t *pCtx, uint32_t ulFormat)
{
    if (!pCtx)
        return 0;
    if (ulFormat > IF_FIR_INPUT_I16)
    {
        set_err(pCtx, "if_fir_set_input_format: unknown format %u", ulFormat);
        return 0;
    }
    const int old = pCtx->in_i16;
    pCtx->in_i16 = (ulFormat == IF_FIR_INPUT_I16);
    // would the requested backend still apply?  (decided before anything is changed)
    const uint32_t b = resolve_backend(pCtx, pCtx->backend_req);
    if (!backend_ok(pCtx, b))
    {
        pCtx->in_i16 = old;
        set_err(pCtx, "if_fir_set_input_format: backend %u does not take this input format", pCtx->backend_req);
        return 0;
    }
    if (old != pCtx->in_i16 && (pCtx->d_fft_tables || pCtx->d_fft_tables_bank)) // the tables carry the sample format's scale
    {
        hipError_t e = hipSetDevice(pCtx->device);
        if (e == hipSuccess)
            e = hipStreamSynchronize(pCtx->stream); // kernels in flight still read them
        if (e != hipSuccess)
        {
            pCtx->in_i16 = old;
            set_err(pCtx, "if_fir_set_input_format: %s", hipGetErrorString(e));
            return 0;
        }
        if (pCtx->d_fft_tables)
            (void)hipFree(pCtx->d_fft_tables);
        if (pCtx->d_fft_tables_bank)
            (void)hipFree(pCtx->d_fft_tables_bank);
        pCtx->d_fft_tables = pCtx->d_fft_tables_bank = nullptr;
    }
    pCtx->backend = b;
    // the history buffer holds samples in the input format: a changed format starts a fresh stream
    if (old != pCtx->in_i16 && !if_fir_reset(pCtx))
        return 0;
    // (a failed upload leaves the tables unbuilt; every launch makes sure of them again, run_device)
    if (b == IF_FIR_BACKEND_HIP_FFT && !ensure_fft_tables(pCtx))
        return 0;
    return 1;
}
