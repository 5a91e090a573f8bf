// fake_rccl.cpp — TEST INFRASTRUCTURE: an in-process stand-in for the eight RCCL entry points if_fir_mc.cpp uses, so
// that the multi-rank path of if_fir_mc_* (chunked scatter / filter / gather, two streams, events, status word) can
// EXECUTE on a one-GPU box with the ranks as threads of one process (RCCL itself refuses two ranks on one device,
// profiles/r02_mc_same_device_attempt.txt).  Loaded through IF_FIR_RCCL_LIBRARY; never shipped, never linked.
//
// Semantics kept from NCCL point-to-point: operations between a pair of ranks match in posting order; a group's
// operations are all posted before any of them is waited for; a send and its receive must carry the same byte count;
// the transfer is ordered behind the work queued before it on the SENDER's stream and the sender's stream continues
// only after the data has been read; the receiver's stream continues only after the data has landed.  Transport = a
// device-to-device copy.  Difference: the host blocks in ncclGroupEnd until every operation of the group has found its
// partner (real RCCL blocks a kernel instead) — the same rendezvous, seen from the host.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

extern "C"
{
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef int ncclDataType_t;
}

namespace
{
struct Op
{
    void *buf;
    size_t bytes;
    hipStream_t stream;
    bool matched = false, bad = false;
};
struct World
{
    int n = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<Op *>> sends, recvs; // key (src, dst), posting order
    long transfers = 0, bytes = 0;
};
struct Comm
{
    std::shared_ptr<World> w;
    int rank;
};
struct Pending
{
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
std::mutex g_m;
std::map<std::string, std::shared_ptr<World>> g_worlds;
unsigned g_ids = 0;
thread_local int t_depth = 0;
thread_local std::vector<Pending> t_pending;

// both sides known: order the copy behind the sender's stream, run it on the receiver's, let the sender go on after it
bool transfer(Op &s, Op &r)
{
    if (s.bytes != r.bytes)
        return false;
    hipEvent_t ready, done;
    bool ok = hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventRecord(ready, s.stream) == hipSuccess && hipStreamWaitEvent(r.stream, ready, 0) == hipSuccess;
    ok = ok && hipMemcpyAsync(r.buf, s.buf, s.bytes, hipMemcpyDeviceToDevice, r.stream) == hipSuccess;
    ok = ok && hipEventRecord(done, r.stream) == hipSuccess && hipStreamWaitEvent(s.stream, done, 0) == hipSuccess;
    (void)hipEventDestroy(ready);
    (void)hipEventDestroy(done);
    return ok;
}

ncclResult_t run(std::vector<Pending> &ops)
{
    if (ops.empty())
        return ncclSuccess;
    World &w = *ops[0].comm->w;
    std::vector<std::unique_ptr<Op>> mine;
    bool bad = false;
    std::unique_lock<std::mutex> lock(w.m);
    for (Pending &p : ops)
    {
        const int me = p.comm->rank;
        if (p.peer < 0 || p.peer >= w.n || p.peer == me)
            return ncclInvalidArgument;
        const std::pair<int, int> key = p.send ? std::make_pair(me, p.peer) : std::make_pair(p.peer, me);
        std::deque<Op *> &other = p.send ? w.recvs[key] : w.sends[key];
        mine.emplace_back(new Op{p.buf, p.bytes, p.stream});
        Op *op = mine.back().get();
        if (!other.empty())
        {
            Op *partner = other.front();
            other.pop_front();
            const bool ok = p.send ? transfer(*op, *partner) : transfer(*partner, *op);
            op->matched = partner->matched = true;
            op->bad = partner->bad = !ok;
            w.transfers++;
            w.bytes += (long)p.bytes;
        }
        else
            (p.send ? w.sends[key] : w.recvs[key]).push_back(op);
    }
    w.cv.notify_all();
    w.cv.wait(lock, [&] { for (auto &o : mine) if (!o->matched) return false; return true; });
    for (auto &o : mine)
        bad = bad || o->bad;
    return bad ? ncclInvalidUsage : ncclSuccess;
}
} // namespace

extern "C"
{
#define FAKE_API __attribute__((visibility("default")))
FAKE_API ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lock(g_m);
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "fake-rccl-world-%u", ++g_ids);
    return ncclSuccess;
}
FAKE_API ncclResult_t ncclCommInitRank(Comm **comm, int n, ncclUniqueId id, int rank)
{
    if (!comm || n < 1 || rank < 0 || rank >= n)
        return ncclInvalidArgument;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lock(g_m);
        std::shared_ptr<World> &slot = g_worlds[std::string(id.internal, sizeof(id.internal))];
        if (!slot)
        {
            slot = std::make_shared<World>();
            slot->n = n;
        }
        w = slot;
    }
    if (w->n != n)
        return ncclInvalidUsage;
    std::unique_lock<std::mutex> lock(w->m);
    w->joined++;
    w->cv.notify_all();
    w->cv.wait(lock, [&] { return w->joined >= w->n; }); // like the real call: returns when every rank has joined
    *comm = new Comm{w, rank};
    return ncclSuccess;
}
FAKE_API ncclResult_t ncclCommDestroy(Comm *comm)
{
    delete comm;
    return ncclSuccess;
}
FAKE_API ncclResult_t ncclCommAbort(Comm *comm)
{
    delete comm;
    return ncclSuccess;
}
FAKE_API ncclResult_t ncclGroupStart()
{
    t_depth++;
    return ncclSuccess;
}
FAKE_API ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0)
        return ncclInvalidUsage;
    if (--t_depth > 0)
        return ncclSuccess;
    std::vector<Pending> ops;
    ops.swap(t_pending);
    return run(ops);
}
static ncclResult_t post(bool send, void *buf, size_t count, ncclDataType_t type, int peer, Comm *comm, hipStream_t stream)
{
    if (!comm || !buf || type != 1 /* ncclUint8: the library moves bytes */)
        return ncclInvalidArgument;
    t_pending.push_back({send, buf, count, peer, comm, stream});
    if (t_depth > 0)
        return ncclSuccess;
    std::vector<Pending> ops;
    ops.swap(t_pending);
    return run(ops);
}
FAKE_API ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, Comm *comm, hipStream_t stream)
{
    return post(true, const_cast<void *>(buf), count, type, peer, comm, stream);
}
FAKE_API ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, Comm *comm, hipStream_t stream)
{
    return post(false, buf, count, type, peer, comm, stream);
}
// asynchronous error state: healthy, unless the test asks one rank to report a failure (FAKE_RCCL_ASYNC_ERROR_RANK=r): lets
// the tests reach the library's polling wait without breaking anything
FAKE_API ncclResult_t ncclCommGetAsyncError(Comm *comm, ncclResult_t *err)
{
    if (!comm || !err)
        return ncclInvalidArgument;
    const char *e = getenv("FAKE_RCCL_ASYNC_ERROR_RANK");
    *err = (e && *e && atoi(e) == comm->rank) ? ncclSystemError : ncclSuccess;
    return ncclSuccess;
}
FAKE_API const char *ncclGetErrorString(ncclResult_t r)
{
    static const char *names[] = {"success", "unhandled device error", "system error", "internal error",
                                  "invalid argument", "invalid usage (byte counts of a send and its receive differ?)"};
    return (int)r >= 0 && (int)r <= 5 ? names[(int)r] : "unknown";
}
// test hook: transfers and bytes moved so far in the world of `comm`
FAKE_API void fake_rccl_stats(Comm *comm, long *transfers, long *bytes)
{
    std::lock_guard<std::mutex> lock(comm->w->m);
    *transfers = comm->w->transfers;
    *bytes = comm->w->bytes;
}
}
