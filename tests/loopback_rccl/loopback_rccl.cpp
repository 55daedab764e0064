// loopback_rccl.cpp -- TEST INFRASTRUCTURE, not product: an in-process stand-in for the nine RCCL
// entry points libpedoni_hip resolves with dlopen (pedoni_amd/csrc/shard.hpp), selected with
// PEDONI_RCCL_LIB=<this .so>.  It lets the multi-rank driver below the C-ABI -- pedoni_shard_tick_n:
// the rank+-1 ncclSend / ncclRecv exchange, the overlapped form on its own stream, the re-cut's
// ncclAllReduce + bulk exchange -- run with world > 1 on ONE GPU: every rank is a host thread of
// one process with its own model and stream, and a "send" is a device copy into a staging buffer
// the matching "receive" copies out of, ordered by HIP events.  What it does NOT test is RCCL
// itself or the wire; what it does test is every line of OUR side of the protocol (which buffer,
// which offset, which peer, which stream, group bracketing, message sizes), which a 1-GPU box
// cannot reach through the real library (RCCL refuses two ranks on one device).
//
// Semantics kept from NCCL: operations inside ncclGroupStart / ncclGroupEnd are issued when the
// outermost group closes, sends before receives (so a rank may post send + receive to the same
// peer in one group without deadlock); a receive must match the size of the message it takes;
// ncclCommInitRank blocks until all ranks of the id have joined.  Every host-side wait is
// bounded (LOOPBACK_RCCL_TIMEOUT_S, default 60 s): a protocol bug fails the test, never hangs it.
//
// Fault injection for tests: LOOPBACK_RCCL_FAIL_SEND=<k> makes the k-th ncclSend of the process
// (0-based) return ncclInternalError; loopback_rccl_group_depth() reports the calling thread's
// open-group depth (must be 0 after any pedoni_shard_* call, failed or not).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Message { void* staging; size_t bytes; hipEvent_t ready; };

struct World {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int joined = 0, destroyed = 0;
    std::map<std::pair<int, int>, std::deque<Message>> box;   // (src, dst) -> messages in order
    std::vector<void*> garbage_mem;
    std::vector<hipEvent_t> garbage_ev;
    // all-reduce rendezvous
    int ar_arrived = 0;
    unsigned long long ar_gen = 0;
    std::vector<uint32_t> ar_sum, ar_result;
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
std::atomic<unsigned> g_next_id{1};
std::atomic<long> g_sends{0};
std::atomic<long> g_stat_send{0}, g_stat_recv{0}, g_stat_allreduce{0};

std::chrono::seconds timeout()
{
    const char* t = std::getenv("LOOPBACK_RCCL_TIMEOUT_S");
    return std::chrono::seconds(t ? std::atoi(t) : 60);
}

} // namespace

struct ncclComm { World* w; int rank; };

namespace {

struct Op { bool send; const void* src; void* dst; size_t bytes; int peer; ncclComm* c; hipStream_t st; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

ncclResult_t do_send(const Op& o)
{
    World* w = o.c->w;
    Message m{nullptr, o.bytes, nullptr};
    if (hipMalloc(&m.staging, o.bytes ? o.bytes : 4) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&m.ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpyAsync(m.staging, o.src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess)
        return ncclUnhandledCudaError;
    if (hipEventRecord(m.ready, o.st) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->box[{o.c->rank, o.peer}].push_back(m);
    }
    w->cv.notify_all();
    g_stat_send++;
    return ncclSuccess;
}

ncclResult_t do_recv(const Op& o)
{
    World* w = o.c->w;
    Message m{};
    {
        std::unique_lock<std::mutex> lk(w->mu);
        auto& q = w->box[{o.peer, o.c->rank}];
        if (!w->cv.wait_for(lk, timeout(), [&] { return !q.empty(); })) return ncclInternalError;   // no matching send
        m = q.front();
        q.pop_front();
        w->garbage_mem.push_back(m.staging);
        w->garbage_ev.push_back(m.ready);
    }
    if (m.bytes != o.bytes) return ncclInvalidArgument;       // size mismatch between the two ends
    if (hipStreamWaitEvent(o.st, m.ready, 0) != hipSuccess) return ncclUnhandledCudaError;
    // LOOPBACK_RCCL_RECV_DELAY_US: a straggling neighbour -- the message reaches the receiver's stream this
    // much later (a host function in stream order: everything behind it on that stream waits)
    static const long delay_us = [] { const char* d = std::getenv("LOOPBACK_RCCL_RECV_DELAY_US"); return d ? std::atol(d) : 0L; }();
    if (delay_us > 0 && o.bytes > 64 &&
        hipLaunchHostFunc(o.st, [](void* us) { std::this_thread::sleep_for(std::chrono::microseconds((long)(intptr_t)us)); },
                          (void*)(intptr_t)delay_us) != hipSuccess)
        return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpyAsync(o.dst, m.staging, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess)
        return ncclUnhandledCudaError;
    g_stat_recv++;
    return ncclSuccess;
}

ncclResult_t flush_ops()
{
    std::vector<Op> ops;
    ops.swap(t_ops);
    ncclResult_t first = ncclSuccess;
    for (const Op& o : ops)
        if (o.send) { ncclResult_t r = do_send(o); if (first == ncclSuccess) first = r; }
    for (const Op& o : ops)
        if (!o.send && first == ncclSuccess) { ncclResult_t r = do_recv(o); if (first == ncclSuccess) first = r; }
    return first;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

} // namespace

extern "C" {

int loopback_rccl_group_depth(void) { return t_depth; }
void loopback_rccl_stats(long out[3]) { out[0] = g_stat_send; out[1] = g_stat_recv; out[2] = g_stat_allreduce; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    const unsigned k = g_next_id++;
    std::memcpy(id->internal, "LOOPBACK", 8);
    std::memcpy(id->internal + 8, &k, sizeof k);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (std::memcmp(id.internal, "LOOPBACK", 8) != 0) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof id.internal);
    World* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) { w = new World(); w->n = nranks; g_worlds[key] = w; }
        else w = it->second;
    }
    if (w->n != nranks) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->joined += 1;
        w->cv.notify_all();
        if (!w->cv.wait_for(lk, timeout(), [&] { return w->joined >= w->n; })) return ncclInternalError;
    }
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    World* w = comm->w;
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        last = ++w->destroyed == w->n;
    }
    if (last) {
        hipDeviceSynchronize();
        for (void* p : w->garbage_mem) hipFree(p);
        for (hipEvent_t e : w->garbage_ev) hipEventDestroy(e);
        for (auto& kv : w->box)
            for (Message& m : kv.second) { hipFree(m.staging); hipEventDestroy(m.ready); }
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto it = g_worlds.begin(); it != g_worlds.end(); ++it)
            if (it->second == w) { g_worlds.erase(it); break; }
        delete w;
    }
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return flush_ops();
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->w->n || !type_bytes(datatype)) return ncclInvalidArgument;
    const char* f = std::getenv("LOOPBACK_RCCL_FAIL_SEND");
    const long k = g_sends++;
    if (f && std::atol(f) == k) return ncclInternalError;
    Op o{true, sendbuff, nullptr, count * type_bytes(datatype), peer, comm, stream};
    if (t_depth > 0) { t_ops.push_back(o); return ncclSuccess; }
    return do_send(o);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->w->n || !type_bytes(datatype)) return ncclInvalidArgument;
    Op o{false, nullptr, recvbuff, count * type_bytes(datatype), peer, comm, stream};
    if (t_depth > 0) { t_ops.push_back(o); return ncclSuccess; }
    return do_recv(o);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                           ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || datatype != ncclUint32 || op != ncclSum) return ncclInvalidArgument;   // all the driver uses
    World* w = comm->w;
    std::vector<uint32_t> mine(count);
    if (hipMemcpyAsync(mine.data(), sendbuff, count * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return ncclUnhandledCudaError;
    std::vector<uint32_t> result;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        if (w->ar_arrived == 0) w->ar_sum.assign(count, 0u);
        if (w->ar_sum.size() != count) return ncclInvalidArgument;
        for (size_t i = 0; i < count; ++i) w->ar_sum[i] += mine[i];
        const unsigned long long gen = w->ar_gen;
        if (++w->ar_arrived == w->n) {
            w->ar_result = w->ar_sum;
            w->ar_arrived = 0;
            w->ar_gen += 1;
            w->cv.notify_all();
        } else if (!w->cv.wait_for(lk, timeout(), [&] { return w->ar_gen != gen; })) {
            return ncclInternalError;
        }
        result = w->ar_result;
    }
    if (hipMemcpyAsync(recvbuff, result.data(), count * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return ncclUnhandledCudaError;
    g_stat_allreduce++;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback: HIP call failed";
    case ncclInternalError: return "loopback: internal error (injected fault, or no matching peer call within the timeout)";
    case ncclInvalidArgument: return "loopback: invalid argument (or send / receive sizes differ)";
    case ncclInvalidUsage: return "loopback: invalid usage (ncclGroupEnd without ncclGroupStart)";
    default: return "loopback: error";
    }
}

} // extern "C"
