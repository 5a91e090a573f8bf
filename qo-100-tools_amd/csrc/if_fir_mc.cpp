// if_fir_mc.cpp — multi-channel front of the C-ABI (include/if_fir.h, if_fir_mc_*): channel c is filtered by rank
// c mod world, one process per GPU.  When the channel inputs live on rank 0 the library moves them itself: one grouped
// batch of RCCL point-to-point sends (root -> owners) before the filters and one (owners -> root) after them, so the
// root drives its xGMI links concurrently.  No reduction, no collective in the filtering itself.
//
// BUILD-DEFINED: the reference has no multi-channel (or any) filter surface.  librccl is opened with
// dlopen() on first use, so single-GPU users of libif_fir.so do not load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "if_fir.h"

#define IF_FIR_API extern "C" __attribute__((visibility("default")))

namespace
{
struct RcclApi
{
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
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
    g_rccl.lib = lib;
    return &g_rccl;
}

thread_local char g_mc_init_err[256] = "";
} // namespace

struct if_fir_mc_ctx
{
    uint32_t channels = 0, taps = 0, decim = 0, rank = 0, world = 1;
    uint32_t in_bytes = 8; // bytes per input sample (8 = float32 I,Q; 4 = int16 I,Q)
    uint64_t max_samples = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<if_fir_ctx_t *> fir; // per channel; nullptr for channels other ranks own
    std::vector<void *> stage_in, stage_out; // owned channels of non-root ranks
    RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    mutable char err[256] = "";
};

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
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
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
    hipError_t e = hipSetDevice(lDevice);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess)
        MC_INIT_FAIL("if_fir_mc_init: device %d: %s", lDevice, hipGetErrorString(e));
    for (uint32_t c = 0; c < ulChannels; c++)
    {
        if (if_fir_mc_owner(c, ulWorld) != ulRank)
            continue;
        if (!if_fir_init(&ctx->fir[c], pfTaps + (size_t)c * ulTaps, ulTaps, ulDecimation, 0, lDevice))
            MC_INIT_FAIL("if_fir_mc_init: channel %u: %s", c, if_fir_last_error(nullptr));
        if (!if_fir_set_stream(ctx->fir[c], ctx->stream))
            MC_INIT_FAIL("if_fir_mc_init: channel %u: %s", c, if_fir_last_error(ctx->fir[c]));
        if (ulRank != 0)
        {
            const size_t out_max = (size_t)((ullMaxSamples + ulDecimation - 1) / ulDecimation + 1) * 8;
            e = hipMalloc(&ctx->stage_in[c], (size_t)ullMaxSamples * 8);
            if (e == hipSuccess)
                e = hipMalloc(&ctx->stage_out[c], out_max);
            if (e != hipSuccess)
                MC_INIT_FAIL("if_fir_mc_init: staging for channel %u: %s", c, hipGetErrorString(e));
        }
    }
    if (ulWorld > 1)
    {
        ctx->api = rccl();
        if (!ctx->api)
            MC_INIT_FAIL("if_fir_mc_init: %s", g_rccl.why);
        ncclUniqueId id;
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

IF_FIR_API uint8_t if_fir_mc_process_device(if_fir_mc_ctx_t *pCtx, const void *const *ppDevIn, void *const *ppDevOut,
                                            uint64_t ullSamples, uint64_t *pullOutSamples)
{
    if (!pCtx)
        return 0;
    if_fir_mc_ctx *ctx = pCtx;
    const bool root = ctx->rank == 0;
    if (pullOutSamples)
        *pullOutSamples = 0;
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
    const size_t in_bytes = (size_t)ullSamples * ctx->in_bytes;
    // ---- scatter: root -> owners, one group (all peers' links busy at once) ---------------------------------------
    if (ctx->world > 1 && in_bytes)
    {
        MC_RCCL(ctx, ctx->api->GroupStart());
        for (uint32_t c = 0; c < ctx->channels; c++)
        {
            const uint32_t owner = if_fir_mc_owner(c, ctx->world);
            if (owner == 0)
                continue;
            if (root)
                MC_RCCL(ctx, ctx->api->Send(ppDevIn[c], in_bytes, ncclUint8, (int)owner, ctx->comm, ctx->stream));
            else if (owner == ctx->rank)
                MC_RCCL(ctx, ctx->api->Recv(ctx->stage_in[c], in_bytes, ncclUint8, 0, ctx->comm, ctx->stream));
        }
        MC_RCCL(ctx, ctx->api->GroupEnd());
    }
    // ---- this rank's channels, back to back on the context's stream -----------------------------------------------
    uint64_t out_samples = 0;
    bool first = true;
    for (uint32_t c = 0; c < ctx->channels; c++)
    {
        if (!ctx->fir[c])
            continue;
        uint64_t m = 0;
        const void *src = root ? ppDevIn[c] : ctx->stage_in[c];
        void *dst = root ? ppDevOut[c] : ctx->stage_out[c];
        if (!if_fir_process_device(ctx->fir[c], src, dst, ullSamples, &m))
        {
            mc_err(ctx, "channel %u: %s", c, if_fir_last_error(ctx->fir[c]));
            return 0;
        }
        if (first)
            out_samples = m;
        else if (m != out_samples)
        {
            mc_err(ctx, "channel %u produced %llu samples, channel before %llu (streams out of step)", c,
                   (unsigned long long)m, (unsigned long long)out_samples);
            return 0;
        }
        first = false;
    }
    // every channel has consumed the same number of samples, so every channel produces the same count; a rank without
    // channels (world > channels) computes it from rank-independent state
    if (first)
        out_samples = 0;
    // ---- gather: owners -> root ------------------------------------------------------------------------------------
    if (ctx->world > 1 && in_bytes)
    {
        // the root needs the count of the remote channels: identical to its own (channel 0 is always the root's)
        const size_t out_bytes = (size_t)out_samples * 8;
        if (out_bytes)
        {
            MC_RCCL(ctx, ctx->api->GroupStart());
            for (uint32_t c = 0; c < ctx->channels; c++)
            {
                const uint32_t owner = if_fir_mc_owner(c, ctx->world);
                if (owner == 0)
                    continue;
                if (root)
                    MC_RCCL(ctx, ctx->api->Recv(ppDevOut[c], out_bytes, ncclUint8, (int)owner, ctx->comm, ctx->stream));
                else if (owner == ctx->rank)
                    MC_RCCL(ctx, ctx->api->Send(ctx->stage_out[c], out_bytes, ncclUint8, 0, ctx->comm, ctx->stream));
            }
            MC_RCCL(ctx, ctx->api->GroupEnd());
        }
    }
    MC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (pullOutSamples)
        *pullOutSamples = out_samples;
    return 1;
}
