// if_fir_mc.cpp — multi-channel front of the C-ABI (include/if_fir.h, if_fir_mc_*): channel c is filtered by rank
// c mod world, one process per GPU.  When the channel inputs live on rank 0 the library moves them itself, in CHUNKS of
// ~2^24 samples: grouped RCCL point-to-point sends root -> owners (the root drives all its xGMI links at once) on a
// transfer stream, the filters of a chunk on a second stream as soon as that chunk has landed, the chunk's outputs back
// to the root behind the next chunk's scatter.  The transfer plan is a pure host function (mc_plan, exported for the
// CPU tests as if_fir_mc_debug_plan): what every rank sends and receives, in which group, in which order.  No
// reduction, no collective in the filtering itself.
//
// BUILD-DEFINED: the reference has no multi-channel (or any) filter surface.  librccl is opened with
// dlopen() on first use, so single-GPU users of libif_fir.so do not load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <new>
#include <vector>

#include "if_fir.h"
#ifdef IF_FIR_DEVELOPMENT
#include "if_fir_debug.h"
#endif

#define IF_FIR_API extern "C" __attribute__((visibility("default")))

namespace
{
struct RcclApi
{
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr; // optional
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[200] = "";
};

std::mutex g_rccl_mutex; // the only global state of the library: the lazily opened RCCL entry points
RcclApi g_rccl;

// returns nullptr (with g_rccl.why set) when librccl cannot be opened
RcclApi *rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib)
        return &g_rccl;
    // IF_FIR_RCCL_LIBRARY: an explicit library path (another RCCL build; the tests load an in-process stand-in for the
    // transport here, tests/c/fake_rccl.cpp, to run several ranks as threads on one GPU)
    const char *names[] = {getenv("IF_FIR_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names)
        if (n && *n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
            break;
    if (!lib)
    {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "cannot open librccl: %s", dlerror());
        return nullptr;
    }
#define IF_FIR_SYM(field, name)                                                          \
    do                                                                                   \
    {                                                                                    \
        g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));       \
        if (!g_rccl.field)                                                               \
        {                                                                                \
            snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks %s", name);          \
            dlclose(lib);                                                                \
            return nullptr;                                                              \
        }                                                                                \
    } while (0)
    IF_FIR_SYM(GetUniqueId, "ncclGetUniqueId");
    IF_FIR_SYM(CommInitRank, "ncclCommInitRank");
    IF_FIR_SYM(CommDestroy, "ncclCommDestroy");
    IF_FIR_SYM(Send, "ncclSend");
    IF_FIR_SYM(Recv, "ncclRecv");
    IF_FIR_SYM(GroupStart, "ncclGroupStart");
    IF_FIR_SYM(GroupEnd, "ncclGroupEnd");
    IF_FIR_SYM(GetErrorString, "ncclGetErrorString");
#undef IF_FIR_SYM
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(lib, "ncclCommAbort"));
    g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(lib, "ncclCommGetAsyncError"));
    g_rccl.lib = lib;
    return &g_rccl;
}

thread_local char g_mc_init_err[256] = "";

// chunks of a call: multiples of THIS filter's block advance and of twice its decimation (mc_chunk_unit, see the transfer plan
// below); the default request is ~2^24 samples
constexpr uint64_t MC_CHUNK_DEFAULT = 16773120; // 78 x 215 040 (round 3's global unit, the lcm of the block advances then)
} // namespace

namespace if_fir
{
bool fft_tail(int T, int D, int *pF, int *pSub); // if_fir_fft.hip
bool fft_odd_tail(int T, int D, int *pF, int *pSub, int *pOvlr);
int fft_block_advance(int T, int D);
}
// internal to the library (if_fir_shim.cpp, hidden): device address of the context's queue fault counter
extern "C" __attribute__((visibility("hidden"))) const uint32_t *if_fir_internal_fault_word(const if_fir_ctx_t *pCtx);

struct if_fir_mc_ctx
{
    uint32_t channels = 0, taps = 0, decim = 0, rank = 0, world = 1;
    // LOOPBACK (development library, IF_FIR_MC_LOOPBACK=N, one rank): this process plays ALL ranks of an N-rank world (2..16) over
    // a one-rank communicator -- every send of the plan is matched by its receive in the same group, peer = itself -- so that
    // the whole protocol (chunks, groups, events, staging slots, status words, the polling wait) runs over the real librccl on
    // a one-GPU box.  Channel c is "rank c mod N's": all but rank 0's are staged, filtered from their slots, gathered back.
    bool loop = false;
    uint32_t vranks = 0;
    uint32_t in_bytes = 8; // bytes per input sample (8 = float32 I,Q; 4 = int16 I,Q)
    uint64_t max_samples = 0;
    uint64_t consumed = 0;      // samples per channel since init/reset (every rank counts: sizes of the gather pieces)
    uint64_t chunk_samples = 0; // requested transfer/filter chunk (mc_effective_chunk rounds it), 0 = whole call in one piece
    double timeout_s = 300.0;   // IF_FIR_MC_TIMEOUT_S, read once at init
    const uint32_t **d_faultp = nullptr; // device: the owned channels' queue fault counters (status word of the last group)
    uint32_t n_faultp = 0;
    int device = 0;
    hipStream_t stream = nullptr;      // filters
    hipStream_t xfer_stream = nullptr; // RCCL transfers
    uint32_t *d_status = nullptr;      // device: [0] this rank's status word, [1 + r] = rank r's as received by the root
    bool comm_broken = false;
    std::vector<if_fir_ctx_t *> fir; // per channel; nullptr for channels other ranks own
    // owned channels of ranks other than 0: TWO slots each (chunk k lives in slot k & 1; the transfer plan's order makes the
    // reuse safe, see if_fir_mc_process_device), sized by the chunk, not by the longest call
    std::vector<void *> stage_in, stage_out;
    uint64_t slot_samples = 0;                  // input samples one slot holds
    size_t slot_in_bytes = 0, slot_out_bytes = 0;
    RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    mutable char err[256] = "";
};

static uint64_t mc_slot_samples(uint64_t samples, uint64_t chunk, uint32_t decim, uint32_t taps);

static void mc_err(const if_fir_mc_ctx *ctx, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_mc_init_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
}

#define MC_HIP(ctx, call)                                                                          \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
        {                                                                                          \
            mc_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 0;                                                                              \
        }                                                                                          \
    } while (0)
#define MC_RCCL(ctx, call)                                                                                  \
    do                                                                                                      \
    {                                                                                                       \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess)                                                                              \
        {                                                                                                   \
            mc_err(ctx, "%s failed: %s (%s:%d)", #call, (ctx)->api->GetErrorString(r_), __FILE__, __LINE__); \
            return 0;                                                                                       \
        }                                                                                                   \
    } while (0)

IF_FIR_API uint32_t if_fir_mc_owner(uint32_t ulChannel, uint32_t ulWorld)
{
    return ulWorld ? ulChannel % ulWorld : 0;
}

IF_FIR_API uint8_t if_fir_mc_unique_id(uint8_t *pubId)
{
    static_assert(IF_FIR_MC_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!pubId)
    {
        mc_err(nullptr, "if_fir_mc_unique_id: NULL buffer");
        return 0;
    }
    RcclApi *api = rccl();
    if (!api)
    {
        mc_err(nullptr, "if_fir_mc_unique_id: %s", g_rccl.why);
        return 0;
    }
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess)
    {
        mc_err(nullptr, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
        return 0;
    }
    memcpy(pubId, id.internal, IF_FIR_MC_ID_BYTES);
    return 1;
}

static void mc_free(if_fir_mc_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->xfer_stream)
        (void)hipStreamSynchronize(ctx->xfer_stream);
    for (if_fir_ctx_t *f : ctx->fir)
        if (f)
            if_fir_destroy(f);
    for (void *p : ctx->stage_in)
        if (p)
            (void)hipFree(p);
    for (void *p : ctx->stage_out)
        if (p)
            (void)hipFree(p);
    if (ctx->comm)
        (void)ctx->api->CommDestroy(ctx->comm);
    if (ctx->d_status)
        (void)hipFree(ctx->d_status);
    if (ctx->d_faultp)
        (void)hipFree(ctx->d_faultp);
    if (ctx->xfer_stream)
        (void)hipStreamDestroy(ctx->xfer_stream);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// staging of the ranks other than 0: two slots of `need` input samples (and their outputs) per owned channel; grows on
// demand (a longer chunk setting), never shrinks
static uint8_t mc_ensure_staging(if_fir_mc_ctx *ctx, uint64_t need)
{
    if ((ctx->rank == 0 && !ctx->loop) || need <= ctx->slot_samples)
        return 1;
    if (hipSetDevice(ctx->device) != hipSuccess)
        return 0;
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->xfer_stream)
        (void)hipStreamSynchronize(ctx->xfer_stream);
    const size_t in_b = ((size_t)need * 8 + 255) & ~(size_t)255;
    const size_t out_b = ((size_t)(need / ctx->decim + 2) * 8 + 255) & ~(size_t)255;
    for (uint32_t c = 0; c < ctx->channels; c++)
    {
        if (!ctx->fir[c] || (ctx->loop && c % ctx->vranks == 0u)) // (loopback: virtual rank 0's own channels are filtered in place)
            continue;
        if (ctx->stage_in[c])
            (void)hipFree(ctx->stage_in[c]);
        if (ctx->stage_out[c])
            (void)hipFree(ctx->stage_out[c]);
        ctx->stage_in[c] = ctx->stage_out[c] = nullptr;
        ctx->slot_samples = 0;
        hipError_t e = hipMalloc(&ctx->stage_in[c], 2 * in_b);
        if (e == hipSuccess)
            e = hipMalloc(&ctx->stage_out[c], 2 * out_b);
        if (e != hipSuccess)
        {
            mc_err(ctx, "staging for channel %u (2 x %llu samples): %s", c, (unsigned long long)need, hipGetErrorString(e));
            return 0;
        }
    }
    ctx->slot_samples = need;
    ctx->slot_in_bytes = in_b;
    ctx->slot_out_bytes = out_b;
    return 1;
}

IF_FIR_API uint8_t if_fir_mc_init(if_fir_mc_ctx_t **ppCtx, uint32_t ulChannels, const float *pfTaps, uint32_t ulTaps,
                                  uint32_t ulDecimation, uint64_t ullMaxSamples, int32_t lDevice, uint32_t ulRank,
                                  uint32_t ulWorld, const uint8_t *pubId)
{
    if (ppCtx)
        *ppCtx = nullptr;
    if (!ppCtx || !pfTaps || !ulChannels || !ulWorld || ulRank >= ulWorld || !ullMaxSamples)
    {
        mc_err(nullptr, "if_fir_mc_init: invalid argument (channels %u, rank %u of %u, max samples %llu)", ulChannels,
               ulRank, ulWorld, (unsigned long long)ullMaxSamples);
        return 0;
    }
    if (ulWorld > 1 && !pubId)
    {
        mc_err(nullptr, "if_fir_mc_init: %u ranks need the unique id of if_fir_mc_unique_id() from rank 0", ulWorld);
        return 0;
    }
    if_fir_mc_ctx *ctx = new (std::nothrow) if_fir_mc_ctx;
    if (!ctx)
    {
        mc_err(nullptr, "if_fir_mc_init: out of memory");
        return 0;
    }
    ctx->channels = ulChannels;
    ctx->taps = ulTaps;
    ctx->decim = ulDecimation;
    ctx->rank = ulRank;
    ctx->world = ulWorld;
    ctx->max_samples = ullMaxSamples;
    ctx->device = lDevice;
    ctx->fir.assign(ulChannels, nullptr);
    ctx->stage_in.assign(ulChannels, nullptr);
    ctx->stage_out.assign(ulChannels, nullptr);
    // every failure below reports through g_mc_init_err and frees what exists
#define MC_INIT_FAIL(...)             \
    do                                \
    {                                 \
        mc_err(nullptr, __VA_ARGS__); \
        mc_free(ctx);                 \
        return 0;                     \
    } while (0)
#ifdef IF_FIR_DEVELOPMENT
    {
        const char *lb = getenv("IF_FIR_MC_LOOPBACK");
        const int nv = lb ? atoi(lb) : 0;
        ctx->vranks = nv == 1 ? 2u : (nv >= 2 && nv <= 16) ? (uint32_t)nv : 0u; // "1" = two virtual ranks
        ctx->loop = ulWorld == 1 && ulChannels >= 2 && ctx->vranks >= 2;
    }
#endif
    const bool transport = ulWorld > 1 || ctx->loop;
    const uint32_t vworld = ctx->loop ? ctx->vranks : ulWorld; // ranks the transfer plan is made for
    hipError_t e = hipSetDevice(lDevice);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess && transport)
        e = hipStreamCreateWithFlags(&ctx->xfer_stream, hipStreamNonBlocking);
    if (e == hipSuccess && transport)
        e = hipMalloc(reinterpret_cast<void **>(&ctx->d_status), (size_t)(1 + vworld) * 4);
    if (e == hipSuccess && transport)
        e = hipMemset(ctx->d_status, 0, (size_t)(1 + vworld) * 4);
    if (e != hipSuccess)
        MC_INIT_FAIL("if_fir_mc_init: device %d: %s", lDevice, hipGetErrorString(e));
    // (one rank moves nothing: its calls are not split unless asked to -- a 2^24-sample piece keeps the overlap-save kernel's
    // 2048 waves busy for two blocks each and runs at half the rate of a whole 2^28-sample call)
    ctx->chunk_samples = transport ? MC_CHUNK_DEFAULT : 0;
    for (uint32_t c = 0; c < ulChannels; c++)
    {
        if (if_fir_mc_owner(c, ulWorld) != ulRank)
            continue;
        if (!if_fir_init(&ctx->fir[c], pfTaps + (size_t)c * ulTaps, ulTaps, ulDecimation, 0, lDevice))
            MC_INIT_FAIL("if_fir_mc_init: channel %u: %s", c, if_fir_last_error(nullptr));
        if (!if_fir_set_stream(ctx->fir[c], ctx->stream))
            MC_INIT_FAIL("if_fir_mc_init: channel %u: %s", c, if_fir_last_error(ctx->fir[c]));
    }
    {
        const char *te = getenv("IF_FIR_MC_TIMEOUT_S");
        if (te && atof(te) > 0)
            ctx->timeout_s = atof(te);
    }
    if (transport)
    {
        // the owned channels' queue fault counters: folded into this rank's status word on the device (mc_status_kernel)
        std::vector<const uint32_t *> fp;
        for (uint32_t c = 0; c < ulChannels; c++)
            if (ctx->fir[c])
                fp.push_back(if_fir_internal_fault_word(ctx->fir[c]));
        ctx->n_faultp = (uint32_t)fp.size();
        if (!fp.empty())
        {
            e = hipMalloc(reinterpret_cast<void **>(&ctx->d_faultp), fp.size() * sizeof(void *));
            if (e == hipSuccess)
                e = hipMemcpy(ctx->d_faultp, fp.data(), fp.size() * sizeof(void *), hipMemcpyHostToDevice);
            if (e != hipSuccess)
                MC_INIT_FAIL("if_fir_mc_init: device %d: %s", lDevice, hipGetErrorString(e));
        }
    }
    if (!mc_ensure_staging(ctx, mc_slot_samples(ullMaxSamples, ctx->chunk_samples, ulDecimation, ulTaps)))
    {
        snprintf(g_mc_init_err, sizeof(g_mc_init_err), "if_fir_mc_init: %s", ctx->err);
        mc_free(ctx);
        return 0;
    }
    if (transport)
    {
        ctx->api = rccl();
        if (!ctx->api)
            MC_INIT_FAIL("if_fir_mc_init: %s", g_rccl.why);
        ncclUniqueId id;
        if (ctx->loop)
        {
            const ncclResult_t ri = ctx->api->GetUniqueId(&id); // a one-rank communicator of its own
            if (ri != ncclSuccess)
                MC_INIT_FAIL("if_fir_mc_init: ncclGetUniqueId: %s", ctx->api->GetErrorString(ri));
        }
        else
            memcpy(id.internal, pubId, IF_FIR_MC_ID_BYTES);
        const ncclResult_t r = ctx->api->CommInitRank(&ctx->comm, (int)ulWorld, id, (int)ulRank);
        if (r != ncclSuccess)
        {
            ctx->comm = nullptr;
            MC_INIT_FAIL("if_fir_mc_init: ncclCommInitRank: %s", ctx->api->GetErrorString(r));
        }
    }
#undef MC_INIT_FAIL
    *ppCtx = ctx;
    return 1;
}

IF_FIR_API void if_fir_mc_destroy(if_fir_mc_ctx_t *pCtx)
{
    mc_free(pCtx);
}

IF_FIR_API const char *if_fir_mc_last_error(const if_fir_mc_ctx_t *pCtx)
{
    return pCtx ? pCtx->err : g_mc_init_err;
}

IF_FIR_API if_fir_ctx_t *if_fir_mc_channel_ctx(if_fir_mc_ctx_t *pCtx, uint32_t ulChannel)
{
    if (!pCtx || ulChannel >= pCtx->channels)
        return nullptr;
    return pCtx->fir[ulChannel];
}

IF_FIR_API uint8_t if_fir_mc_reset(if_fir_mc_ctx_t *pCtx)
{
    if (!pCtx)
        return 0;
    for (uint32_t c = 0; c < pCtx->channels; c++)
        if (pCtx->fir[c] && !if_fir_reset(pCtx->fir[c]))
        {
            mc_err(pCtx, "channel %u: %s", c, if_fir_last_error(pCtx->fir[c]));
            return 0;
        }
    pCtx->consumed = 0;
    return 1;
}

IF_FIR_API uint8_t if_fir_mc_set_input_format(if_fir_mc_ctx_t *pCtx, uint32_t ulFormat)
{
    if (!pCtx)
        return 0;
    if (ulFormat != IF_FIR_INPUT_F32 && ulFormat != IF_FIR_INPUT_I16)
    {
        mc_err(pCtx, "if_fir_mc_set_input_format: unknown format %u", ulFormat);
        return 0;
    }
    for (uint32_t c = 0; c < pCtx->channels; c++)
        if (pCtx->fir[c] && !if_fir_set_input_format(pCtx->fir[c], ulFormat))
        {
            mc_err(pCtx, "channel %u: %s", c, if_fir_last_error(pCtx->fir[c]));
            return 0;
        }
    pCtx->in_bytes = ulFormat == IF_FIR_INPUT_I16 ? 4 : 8;
    return 1;
}

// ---- transfer plan (pure host logic) ----------------------------------------------------------------------------
// A call moves `samples` input samples per channel in chunks.  Chunk k of a call: every remote channel's input piece
// root -> owner (scatter group k), the owners filter it, the output piece owner -> root (gather group k).  Order of
// the groups on every rank's transfer stream: S0, S1, G0, S2, G1, ..., S(n-1), G(n-2), G(n-1), STATUS: the scatter of
// the next chunk is queued ahead of the gather of this one, so the links stay busy while the owners filter.  Inside a
// group the operations between a pair of ranks are posted in channel order on both sides (RCCL matches them in order).
// STATUS: every rank that owns a channel sends one 4-byte word (0 = its filters succeeded) to the root.
// A chunk is a multiple of the block advance of THIS filter's overlap-save kernel (3840, 3584, 3072, 2048 or 1024 samples,
// if_fir::fft_block_advance) so that the blocks of a chunked call start at the same stream positions as those of an unchunked
// one: bit-identical.  (Round 3 used one global unit, the lcm of all block advances.)

enum : uint32_t
{
    MC_SEND = 0,
    MC_RECV = 1,
    MC_PHASE_SCATTER = 0,
    MC_PHASE_GATHER = 1,
    MC_PHASE_STATUS = 2
};
struct McXfer
{
    uint32_t kind, phase, group, peer, channel, chunk;
    uint64_t offset, bytes; // byte offset inside the channel's input (scatter) or output (gather) buffer
    uint32_t as_rank = 0;   // the rank whose operation this is (loopback: one process posts both ranks' operations)
};
struct McChunk
{
    uint64_t in_first, in_count, out_first, out_count; // samples
};

static uint64_t mc_out_count(uint64_t consumed, uint64_t n, uint32_t d)
{
    const uint64_t n0 = (d - consumed % d) % d;
    return n > n0 ? (n - n0 + d - 1) / d : 0;
}

static uint64_t mc_gcd(uint64_t a, uint64_t b)
{
    while (b)
    {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// The decimating tails of the overlap-save kernel (if_fir::fft_tail: decimation 2, 4, 8, 16 and the multiples of 4, 8, 16 that keep
// every sub-th tail output; <= 3073 taps) anchor their block grid at the call's first OUTPUT (n0 samples into the call when the
// stream position is off-phase); every other kernel anchors it at the call's first sample.
static bool mc_grid_follows_phase(uint32_t taps, uint32_t decim)
{
    // (round 4: the odd-decimation kernel -- decimation 3, 9, 15, ... -- anchors its blocks at the first output as well)
    return if_fir::fft_tail((int)taps, (int)decim, nullptr, nullptr) ||
           if_fir::fft_odd_tail((int)taps, (int)decim, nullptr, nullptr, nullptr);
}

// The unit of a context's chunks: lcm(block advance of its filter, 2 D).  Every chunk then produces an even number of outputs
// whatever the phase (the output pieces start at even sample offsets) and starts on the block grid of the unsplit call.
static uint64_t mc_chunk_unit(uint32_t taps, uint32_t decim)
{
    const uint64_t adv = (uint64_t)if_fir::fft_block_advance((int)taps, (int)decim), two_d = 2ull * decim;
    return adv / mc_gcd(adv, two_d) * two_d;
}
// The effective chunk of a request: the multiple of the unit nearest to it (at least one unit).  Round 3 took the lcm of the
// REQUEST and 2 D, which multiplied the default 2^24-sample chunk by 11 .. 61 for decimations with a prime factor the request
// lacked: calls were then never split and the staging slots doubled.
static uint64_t mc_effective_chunk(uint64_t request, uint32_t taps, uint32_t decim)
{
    if (!request)
        return 0;
    const uint64_t u = mc_chunk_unit(taps, decim);
    const uint64_t k = (request + u / 2) / u;
    return (k ? k : 1) * u;
}

// Chunk table of a call (`chunk`: the request, rounded by mc_effective_chunk).  Where the kernel's block grid follows the decimation phase
// the FIRST chunk is n0 samples longer: the chunks behind it then start on phase 0 AND on the block grid of the unsplit
// call, and no block of a chunk reaches past the chunk's end -- chunked results equal unchunked ones bit for bit at any
// phase (the input pieces may then start at odd sample offsets, which the overlap-save kernel accepts).
static void mc_chunks(uint64_t samples, uint64_t chunk, uint64_t consumed, uint32_t decim, uint32_t taps,
                      std::vector<McChunk> &out)
{
    out.clear();
    chunk = mc_effective_chunk(chunk, taps, decim);
    const uint64_t n0 = (decim - consumed % decim) % decim;
    const uint64_t shift = mc_grid_follows_phase(taps, decim) ? n0 : 0;
    uint64_t done = 0, outs = 0;
    while (done < samples)
    {
        const uint64_t want = chunk ? chunk + (done == 0 ? shift : 0) : 0;
        const uint64_t n = (want && samples - done > want) ? want : samples - done;
        const uint64_t m = mc_out_count(consumed + done, n, decim);
        out.push_back({done, n, outs, m});
        done += n;
        outs += m;
    }
}

// capacity (samples) of one staging slot of a rank other than 0: the longest chunk a call of `samples` can have
static uint64_t mc_slot_samples(uint64_t samples, uint64_t chunk, uint32_t decim, uint32_t taps)
{
    if (!chunk)
        return samples;
    const uint64_t eff = mc_effective_chunk(chunk, taps, decim) + decim; // (+ the phase shift of an off-phase first chunk)
    return eff < samples ? eff : samples;
}

// every transfer operation of `rank` for one call, in the order it posts them
static void mc_plan(uint32_t world, uint32_t channels, uint32_t rank, const std::vector<McChunk> &chunks, uint32_t in_bytes,
                    std::vector<McXfer> &ops)
{
    ops.clear();
    if (world < 2 || chunks.empty())
        return;
    const uint32_t n = (uint32_t)chunks.size();
    uint32_t group = 0;
    auto scatter = [&](uint32_t k) {
        for (uint32_t c = 0; c < channels; c++)
        {
            const uint32_t owner = c % world;
            if (owner == 0 || !chunks[k].in_count)
                continue;
            if (rank == 0)
                ops.push_back({MC_SEND, MC_PHASE_SCATTER, group, owner, c, k, chunks[k].in_first * in_bytes, chunks[k].in_count * in_bytes});
            else if (rank == owner)
                ops.push_back({MC_RECV, MC_PHASE_SCATTER, group, 0, c, k, chunks[k].in_first * in_bytes, chunks[k].in_count * in_bytes});
        }
        group++;
    };
    auto gather = [&](uint32_t k) {
        for (uint32_t c = 0; c < channels; c++)
        {
            const uint32_t owner = c % world;
            if (owner == 0 || !chunks[k].out_count)
                continue;
            if (rank == 0)
                ops.push_back({MC_RECV, MC_PHASE_GATHER, group, owner, c, k, chunks[k].out_first * 8, chunks[k].out_count * 8});
            else if (rank == owner)
                ops.push_back({MC_SEND, MC_PHASE_GATHER, group, 0, c, k, chunks[k].out_first * 8, chunks[k].out_count * 8});
        }
        group++;
    };
    scatter(0);
    for (uint32_t k = 1; k < n; k++)
    {
        scatter(k);
        gather(k - 1);
    }
    gather(n - 1);
    for (uint32_t r = 1; r < world && r < channels; r++) // ranks 1 .. min(world, channels) - 1 own at least one channel
    {
        if (rank == 0)
            ops.push_back({MC_RECV, MC_PHASE_STATUS, group, r, 0, 0, (uint64_t)(1 + r) * 4, 4});
        else if (rank == r)
            ops.push_back({MC_SEND, MC_PHASE_STATUS, group, 0, 0, 0, 0, 4});
    }
}

#ifdef IF_FIR_DEVELOPMENT
// Host-only: the transfer plan of one rank as 8 uint64 per operation {kind, phase, group, peer, channel, chunk, offset,
// bytes}; returns the number of operations (also when pullOut is too small or NULL).  ullChunk = 0: one piece.
IF_FIR_API uint32_t if_fir_mc_debug_plan(uint32_t ulWorld, uint32_t ulChannels, uint32_t ulRank, uint64_t ullSamples,
                                         uint32_t ulInBytes, uint32_t ulTaps, uint32_t ulDecimation, uint64_t ullConsumed,
                                         uint64_t ullChunk, uint64_t *pullOut, uint32_t ulMaxOps)
{
    if (!ulWorld || !ulChannels || ulRank >= ulWorld || !ulDecimation || !ulTaps || (ulInBytes != 4 && ulInBytes != 8))
        return 0;
    std::vector<McChunk> chunks;
    mc_chunks(ullSamples, ullChunk, ullConsumed, ulDecimation, ulTaps, chunks);
    std::vector<McXfer> ops;
    mc_plan(ulWorld, ulChannels, ulRank, chunks, ulInBytes, ops);
    for (size_t i = 0; pullOut && i < ops.size() && i < ulMaxOps; i++)
    {
        const McXfer &o = ops[i];
        const uint64_t row[8] = {o.kind, o.phase, o.group, o.peer, o.channel, o.chunk, o.offset, o.bytes};
        memcpy(pullOut + 8 * i, row, sizeof(row));
    }
    return (uint32_t)ops.size();
}

#endif // IF_FIR_DEVELOPMENT

// chunk length of the calls that follow: 0 = default (~2^24 samples), UINT64_MAX = never split, otherwise a request that is
// rounded to the nearest multiple of the context's unit (if_fir_mc_get_chunk_samples reports what is used).  Every rank must
// make the same call.
IF_FIR_API uint8_t if_fir_mc_set_chunk_samples(if_fir_mc_ctx_t *pCtx, uint64_t ullChunk)
{
    if (!pCtx)
        return 0;
    if (ullChunk == UINT64_MAX)
        pCtx->chunk_samples = 0;
    else if (ullChunk == 0)
        pCtx->chunk_samples = (pCtx->world > 1 || pCtx->loop) ? MC_CHUNK_DEFAULT : 0;
    else
        pCtx->chunk_samples = ullChunk;
    return 1;
}

// the chunk in effect (0 = calls are not split) and the unit it is a multiple of
IF_FIR_API uint8_t if_fir_mc_get_chunk_samples(const if_fir_mc_ctx_t *pCtx, uint64_t *pullChunk, uint64_t *pullUnit)
{
    if (!pCtx)
        return 0;
    if (pullChunk)
        *pullChunk = mc_effective_chunk(pCtx->chunk_samples, pCtx->taps, pCtx->decim);
    if (pullUnit)
        *pullUnit = mc_chunk_unit(pCtx->taps, pCtx->decim);
    return 1;
}

// this rank's status word, formed on the device behind the last filter: 1 if the host saw a filter call fail or if any owned
// channel's block queue counted an expired wait during this call (the root must not gather incomplete outputs
// with status 0)
__global__ void mc_status_kernel(uint32_t *status, const uint32_t *const *faults, uint32_t n, uint32_t host_fail)
{
    uint32_t bad = host_fail;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        bad |= (*faults[i] != 0u) ? 1u : 0u;
    bad = __any(bad) ? 1u : 0u;
    if (threadIdx.x == 0)
        status[0] = bad;
}

IF_FIR_API uint8_t if_fir_mc_process_device(if_fir_mc_ctx_t *pCtx, const void *const *ppDevIn, void *const *ppDevOut,
                                            uint64_t ullSamples, uint64_t *pullOutSamples)
{
    if (!pCtx)
        return 0;
    if_fir_mc_ctx *ctx = pCtx;
    const bool root = ctx->rank == 0;
    if (pullOutSamples)
        *pullOutSamples = 0;
    if (ctx->comm_broken)
    {
        mc_err(ctx, "if_fir_mc_process_device: the communicator was aborted after an earlier RCCL failure; create a new context");
        return 0;
    }
    if (ullSamples > ctx->max_samples)
    {
        mc_err(ctx, "if_fir_mc_process_device: %llu samples exceed the %llu of init", (unsigned long long)ullSamples,
               (unsigned long long)ctx->max_samples);
        return 0;
    }
    if (root && (!ppDevIn || !ppDevOut))
    {
        mc_err(ctx, "if_fir_mc_process_device: rank 0 must pass the channel pointer arrays");
        return 0;
    }
    if (root)
        for (uint32_t c = 0; c < ctx->channels; c++)
            if (ullSamples && (!ppDevIn[c] || !ppDevOut[c]))
            {
                mc_err(ctx, "if_fir_mc_process_device: channel %u: NULL device pointer", c);
                return 0;
            }
    MC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<McChunk> chunks;
    mc_chunks(ullSamples, ctx->chunk_samples, ctx->consumed, ctx->decim, ctx->taps, chunks);
    if (chunks.size() > 1)
    {
        // The chunk table (block-grid anchoring, phase shift of the first chunk, sample-aligned piece pointers) is the
        // overlap-save backend's.  A channel switched to another backend through if_fir_mc_channel_ctx() + if_fir_set_backend()
        // would be handed pieces its kernels refuse (16-byte alignment) on a grid that is not theirs: refused here,
        // before anything is posted, on every rank that owns such a channel.  (All ranks run the same configuration.)
        for (uint32_t c = 0; c < ctx->channels; c++)
        {
            if (ctx->fir[c] && if_fir_get_backend(ctx->fir[c]) != IF_FIR_BACKEND_HIP_FFT)
            {
                mc_err(ctx, "if_fir_mc_process_device: channel %u is not on the overlap-save backend: calls are split into chunks on "
                            "that backend only (if_fir_mc_set_chunk_samples(UINT64_MAX) = never split, on every rank)", c);
                return 0;
            }
        }
    }
    {
        uint64_t longest = 0;
        for (const McChunk &ch : chunks)
            longest = ch.in_count > longest ? ch.in_count : longest;
        if (!mc_ensure_staging(ctx, longest))
            return 0;
    }
    // ranks other than 0 keep chunk k of an owned channel in slot k & 1 of its staging buffers.  Reuse is ordered by the
    // plan itself: on the transfer stream S(k+2) comes behind G(k), which waits for filter k (ev_out[k]); filter k+2 waits
    // for S(k+2) (ev_in[k+2]), i.e. for G(k) to have sent slot k & 1 of the outputs.
    auto stage_in_at = [&](uint32_t c, uint32_t k) { return (char *)ctx->stage_in[c] + (size_t)(k & 1u) * ctx->slot_in_bytes; };
    auto stage_out_at = [&](uint32_t c, uint32_t k) { return (char *)ctx->stage_out[c] + (size_t)(k & 1u) * ctx->slot_out_bytes; };
    std::vector<McXfer> ops;
    if (!ctx->loop)
    {
        mc_plan(ctx->world, ctx->channels, ctx->rank, chunks, ctx->in_bytes, ops);
        for (McXfer &o : ops)
            o.as_rank = ctx->rank;
    }
    else
    {
        // loopback: every rank's operations of the virtual world, group by group (rank 0's send and the owner's receive of a
        // transfer land in the same group), every peer = this one rank
        // RCCL matches the sends and receives between one pair of peers in the order they are posted, and here every
        // operation has the same pair (self, self): within a group the operations are put into transfer order -- channel
        // by channel (status words: owner by owner) -- so that the k-th send meets the k-th receive
        for (uint32_t r = 0; r < ctx->vranks; r++)
        {
            std::vector<McXfer> mine;
            mc_plan(ctx->vranks, ctx->channels, r, chunks, ctx->in_bytes, mine);
            for (McXfer &o : mine)
            {
                o.as_rank = r;
                ops.push_back(o);
            }
        }
        auto remote = [](const McXfer &o) { return o.as_rank == 0 ? o.peer : o.as_rank; }; // the rank other than 0 of a transfer
        std::stable_sort(ops.begin(), ops.end(), [&](const McXfer &a, const McXfer &b) {
            if (a.group != b.group)
                return a.group < b.group;
            if (a.channel != b.channel)
                return a.channel < b.channel;
            return remote(a) < remote(b);
        });
        for (McXfer &o : ops)
            o.peer = 0;
    }
    // a channel whose data goes through this process's staging slots: every owned channel of a rank other than 0
    // (loopback: the channels of the virtual ranks other than 0)
    auto staged = [&](uint32_t c) { return ctx->loop ? c % ctx->vranks != 0u : !root; };
    const bool has_staged = ctx->loop || !root;
    const uint32_t nchunks = (uint32_t)chunks.size();
    // events: chunk k's input has landed (transfer stream -> filter stream), chunk k is filtered (filter -> transfer)
    std::vector<hipEvent_t> ev_in(nchunks, nullptr), ev_out(nchunks, nullptr);
    auto free_events = [&]() {
        for (hipEvent_t e : ev_in)
            if (e)
                (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_out)
            if (e)
                (void)hipEventDestroy(e);
    };
    bool local_fail = false, rccl_fail = false; // a filter call failed / an RCCL call failed (text is in ctx->err)
    size_t next_op = 0;
    // posts every operation of the next group; a failing call does not leave the group open
    auto post_group = [&](uint32_t group) {
        if (next_op >= ops.size() || ops[next_op].group != group)
            return; // nothing of this rank in this group
        ncclResult_t first_bad = ctx->api->GroupStart();
        bool opened = first_bad == ncclSuccess;
        for (; next_op < ops.size() && ops[next_op].group == group; next_op++)
        {
            const McXfer &o = ops[next_op];
            if (!opened || rccl_fail)
                continue;
            // (the plan's byte offsets are positions in rank 0's channel buffers; the other ranks use their chunk slots)
            char *buf;
            if (o.phase == MC_PHASE_STATUS)
                buf = reinterpret_cast<char *>(ctx->d_status) + o.offset;
            else if (o.phase == MC_PHASE_SCATTER)
                buf = o.as_rank == 0 ? (char *)const_cast<void *>(ppDevIn[o.channel]) + o.offset : stage_in_at(o.channel, o.chunk);
            else
                buf = o.as_rank == 0 ? (char *)ppDevOut[o.channel] + o.offset : stage_out_at(o.channel, o.chunk);
            const ncclResult_t r = o.kind == MC_SEND
                                       ? ctx->api->Send(buf, o.bytes, ncclUint8, (int)o.peer, ctx->comm, ctx->xfer_stream)
                                       : ctx->api->Recv(buf, o.bytes, ncclUint8, (int)o.peer, ctx->comm, ctx->xfer_stream);
            if (r != ncclSuccess && first_bad == ncclSuccess)
                first_bad = r;
        }
        if (opened)
        {
            const ncclResult_t r = ctx->api->GroupEnd();
            if (r != ncclSuccess && first_bad == ncclSuccess)
                first_bad = r;
        }
        if (first_bad != ncclSuccess && !rccl_fail)
        {
            rccl_fail = true;
            mc_err(ctx, "RCCL transfer group %u failed: %s", group, ctx->api->GetErrorString(first_bad));
        }
    };
    auto filter_chunk = [&](uint32_t k) {
        const McChunk &ch = chunks[k];
        for (uint32_t c = 0; c < ctx->channels && !local_fail; c++)
        {
            if (!ctx->fir[c])
                continue;
            uint64_t m = 0;
            const char *src = !staged(c) ? (const char *)ppDevIn[c] + ch.in_first * ctx->in_bytes : stage_in_at(c, k);
            char *dst = !staged(c) ? (char *)ppDevOut[c] + ch.out_first * 8 : stage_out_at(c, k);
            if (!if_fir_process_device(ctx->fir[c], src, dst, ch.in_count, &m))
            {
                mc_err(ctx, "channel %u: %s", c, if_fir_last_error(ctx->fir[c]));
                local_fail = true;
            }
            else if (m != ch.out_count)
            {
                mc_err(ctx, "channel %u produced %llu samples where the plan has %llu (streams out of step)", c,
                       (unsigned long long)m, (unsigned long long)ch.out_count);
                local_fail = true;
            }
        }
    };
    hipError_t he = hipSuccess;
#define MC_STEP(call)                         \
    do                                        \
    {                                         \
        if (he == hipSuccess)                 \
            he = (call);                      \
    } while (0)
    const bool moving = (ctx->world > 1 || ctx->loop) && !ops.empty();
    uint32_t group = 0;
    auto scatter_step = [&](uint32_t k) {
        if (moving)
        {
            post_group(group);
            if (has_staged)
            {
                MC_STEP(hipEventCreateWithFlags(&ev_in[k], hipEventDisableTiming));
                MC_STEP(hipEventRecord(ev_in[k], ctx->xfer_stream));
                MC_STEP(hipStreamWaitEvent(ctx->stream, ev_in[k], 0)); // the filters of chunk k wait for its input
            }
        }
        group++;
        if (he == hipSuccess && !local_fail)
            filter_chunk(k);
        if (moving && has_staged)
        {
            MC_STEP(hipEventCreateWithFlags(&ev_out[k], hipEventDisableTiming));
            MC_STEP(hipEventRecord(ev_out[k], ctx->stream));
        }
    };
    auto gather_step = [&](uint32_t k) {
        if (moving)
        {
            if (has_staged && ev_out[k])
                MC_STEP(hipStreamWaitEvent(ctx->xfer_stream, ev_out[k], 0)); // send chunk k's outputs once they exist
            post_group(group);
        }
        group++;
    };
    if (nchunks)
    {
        scatter_step(0);
        for (uint32_t k = 1; k < nchunks; k++)
        {
            scatter_step(k);
            gather_step(k - 1);
        }
        gather_step(nchunks - 1);
        if (moving)
        {
            // status word: the protocol above is completed even after a local failure, so that no peer is left waiting;
            // the root learns about it here
            // (the transfer stream is behind the last gather here, which waited for the last filter: the fault counters are final)
            if (has_staged && he == hipSuccess)
            {
                hipLaunchKernelGGL(mc_status_kernel, dim3(1), dim3(64), 0, ctx->xfer_stream, ctx->d_status, ctx->d_faultp, ctx->n_faultp,
                                   local_fail ? 1u : 0u);
                he = hipGetLastError();
            }
            post_group(group);
        }
    }
#undef MC_STEP
    // Wait for both streams.  With ranks on other GPUs a failure THERE (a peer that aborted its communicator, a dead
    // process) shows up here as a transfer that never completes: the wait polls the communicator's asynchronous error
    // state and gives up after IF_FIR_MC_TIMEOUT_S seconds (default 300) instead of blocking for ever.
    auto wait_stream = [&](hipStream_t st) -> hipError_t {
        if (!moving || !ctx->api)
            return hipStreamSynchronize(st);
        const double limit = ctx->timeout_s;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++)
        {
            const hipError_t q = hipStreamQuery(st);
            // (the communicator's error state is looked at once before the first query result is taken: a reported error is
            // never missed; an idle stream with a clean communicator returns without sleeping)
            if (q != hipErrorNotReady && spins > 0)
                return q;
            if ((spins & 63u) == 0u)
            {
                ncclResult_t ar = ncclSuccess;
                if (ctx->api->CommGetAsyncError && ctx->comm &&
                    ctx->api->CommGetAsyncError(ctx->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                {
                    if (!rccl_fail)
                        mc_err(ctx, "RCCL reported an asynchronous error while transfers were in flight: %s",
                               ctx->api->GetErrorString(ar));
                    rccl_fail = true;
                    return hipSuccess;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
                {
                    if (!rccl_fail)
                        mc_err(ctx, "transfers did not complete within %.0f s (a peer has failed or left): communicator aborted",
                               limit);
                    rccl_fail = true;
                    return hipSuccess;
                }
            }
            if (q != hipErrorNotReady)
                return q; // spin 0: the stream is idle and the communicator clean
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    };
    hipError_t se = wait_stream(ctx->stream);
    if (he == hipSuccess)
        he = se;
    if (ctx->xfer_stream && !rccl_fail)
    {
        se = wait_stream(ctx->xfer_stream);
        if (he == hipSuccess)
            he = se;
    }
    if (rccl_fail)
    {
        // abort BEFORE anything waits on the streams again (destroying the events, freeing buffers): the aborted
        // communicator's kernels leave the streams
        if (ctx->api && ctx->api->CommAbort && ctx->comm)
        {
            (void)ctx->api->CommAbort(ctx->comm);
            ctx->comm = nullptr;
        }
        ctx->comm_broken = true;
    }
    free_events();
    if (rccl_fail)
        return 0; // (peers see the aborted communicator through their own polling wait, or run into its time limit)
    if (he != hipSuccess)
    {
        mc_err(ctx, "if_fir_mc_process_device: %s", hipGetErrorString(he));
        return 0;
    }
    if (local_fail)
        return 0;
    // the block queue's fault counters of the owned channels (an expired bounded wait must not pass silently
    // through the multi-channel front either; the streams are idle here, this is one 4-byte read per channel)
    for (uint32_t c = 0; c < ctx->channels; c++)
        if (ctx->fir[c] && !if_fir_synchronize(ctx->fir[c]))
        {
            mc_err(ctx, "channel %u: %s", c, if_fir_last_error(ctx->fir[c]));
            return 0;
        }
    if (root && moving)
    {
        const uint32_t vworld = ctx->loop ? ctx->vranks : ctx->world;
        std::vector<uint32_t> st(1 + vworld, 0);
        MC_HIP(ctx, hipMemcpy(st.data(), ctx->d_status, st.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t r = 1; r < vworld && r < ctx->channels; r++)
            if (st[1 + r])
            {
                mc_err(ctx, "if_fir_mc_process_device: rank %u reported a filter failure (its outputs are invalid)", r);
                return 0;
            }
    }
    ctx->consumed += ullSamples;
    if (pullOutSamples)
        *pullOutSamples = chunks.empty() ? 0 : chunks.back().out_first + chunks.back().out_count;
    return 1;
}
