// shard.hpp -- the multi-GPU driver below the C-ABI (include/pedoni_hip.h, pedoni_shard_*).
// Included at the end of pedoni_hip.hip: it works on PedoniModel's internals (stream, SoA
// buffers, band) through the same functions the single-GPU entry points use.
//
// No reference counterpart (the reference is single-process).  One band
// of neighbor-grid rows per rank.  Per tick, on the model's stream, with no host sync:
//     exchange  ncclGroupStart { ncclSend / ncclRecv with rank-1 and rank+1 } ncclGroupEnd
//               (the band's DOWN list to the band below, its UP list to the band above;
//               fixed-capacity lists, ~0.1 MB: latency-bound, 2 of the 7 xGMI links)
//     halo_unpack -> sort/despawn -> update_states -> halo_pack      (pedoni_hip_halo_tick)
// Every `rebalance_every` ticks the bands are re-cut from the global per-row agent counts
// (ncclAllReduce of a rows-long histogram): the rows that change owner travel to the
// neighbour with their full state in one more grouped send / receive, right after a
// sort/despawn pass, and are re-sorted there -- rows move whole, lower bands keep the lower
// global indices, so the result stays bit-identical to one GPU.
//
// librccl is resolved with dlopen at first use: the library loads (and every single-GPU entry
// point works) where RCCL is absent, and inside a PyTorch process the already loaded
// librccl.so.1 is shared.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "rccl_group.hpp"

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

// One-time resolution (std::call_once: the models are driven from several host threads).
// PEDONI_RCCL_LIB names the library instead of the default search list (a site's own build of
// RCCL; the in-process loop-back transport of tests/loopback_rccl; a path that does not exist
// exercises the "RCCL absent" branch).
RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char* over = std::getenv("PEDONI_RCCL_LIB"); over && *over) names.emplace_back(over);
        else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string last = "?";
        for (const std::string& name : names) {
            api.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
            const char* e = dlerror();           // (read ONCE: dlerror() clears the message it returns)
            if (e) last = e;
        }
        if (!api.handle) {
            api.error = "librccl not found: " + last;
            return;
        }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(api.handle, n);
            if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + n;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return api;
}

std::string rccl_error(const char* what, ncclResult_t r)
{
    return std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
}

#define NCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) return fail(PEDONI_E_HIP, rccl_error(#expr, r_));             \
    } while (0)

// One grouped exchange.  `body(op)` issues the sends / receives through op(call, what); after the
// first failure the remaining calls are skipped, but ncclGroupEnd is ALWAYS called: a rank that
// returned from inside an open group would leave it open for every later call on this thread
// while its peers wait in theirs (rccl_group.hpp has the logic, tested on the CPU with a mock).
template <typename Body> int rccl_group(const char* what, Body&& body)
{
    RcclApi& api = rccl();
    const GroupOutcome<ncclResult_t> o = run_group<ncclResult_t>(
        ncclSuccess, [&] { return api.GroupStart(); }, [&] { return api.GroupEnd(); }, body);
    if (o.ok) return PEDONI_OK;
    return fail(PEDONI_E_HIP, std::string(what) + ": " + rccl_error(o.where, o.code));
}

// ---- device side of the re-cut ---------------------------------------------------------------
// agents per OWNED grid row (zero elsewhere), read off cell_start
__global__ void row_hist_kernel(const uint32_t* __restrict__ cs, int32_t cols, int32_t lo, int32_t hi,
                                int32_t n_rows, uint32_t* __restrict__ hist)
{
    int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n_rows) return;
    hist[r] = (r >= lo && r < hi) ? cs[(int64_t)(r + 1) * cols] - cs[(int64_t)r * cols] : 0u;
}

// whole rows [row_a, row_b) of the sorted order -- one contiguous index range -- as a record
// list (same record as the halo lists); header {count, overflow flag, 0, 0}
__global__ void bulk_pack_kernel(const float2* __restrict__ pos, const float4* __restrict__ velx,
                                 const uint32_t* __restrict__ dest,
                                 const uint32_t* __restrict__ cs, int32_t cols, int32_t row_a, int32_t row_b,
                                 uint32_t cap, uint32_t* __restrict__ out)
{
    const uint32_t begin = row_b > row_a ? cs[(int64_t)row_a * cols] : 0u;
    const uint32_t end = row_b > row_a ? cs[(int64_t)row_b * cols] : 0u;
    const uint32_t n = end - begin;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        out[0] = min(n, cap);
        out[1] = n > cap ? 1u : 0u;
        out[2] = out[3] = 0u;
    }
    if (t >= n || t >= cap) return;
    const uint32_t i = begin + t;
    uint32_t* r = out + PEDONI_HALO_HEADER_WORDS + (size_t)t * PEDONI_HALO_RECORD_WORDS;
    const float2 p = pos[i];
    const float4 v = velx[i];
    r[0] = __float_as_uint(p.x); r[1] = __float_as_uint(p.y);
    r[2] = __float_as_uint(v.x); r[3] = __float_as_uint(v.y);
    r[4] = __float_as_uint(v.w); r[5] = dest[i];
}

// appends both incoming lists behind everything stored: [at0, at0 + nA) then [.., + nB)
__global__ void bulk_unpack_kernel(const uint32_t* __restrict__ in_a, const uint32_t* __restrict__ in_b,
                                   uint32_t cap, uint32_t at0, float2* __restrict__ pos,
                                   float4* __restrict__ velx, uint32_t* __restrict__ dest,
                                   HaloIn* __restrict__ halo)
{
    const uint32_t n_a = min(in_a[0], cap), n_b = min(in_b[0], cap);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        halo->n_above = n_a + n_b;          // the appended range, counted on the device
        halo->counted = 1;
        if (in_a[1] | in_b[1]) atomicOr(&halo->error, 1u);
    }
    const uint32_t* src;
    uint32_t at;
    if (t < cap) {
        if (t >= n_a) return;
        src = in_a + PEDONI_HALO_HEADER_WORDS + (size_t)t * PEDONI_HALO_RECORD_WORDS;
        at = at0 + t;
    } else {
        const uint32_t k = t - cap;
        if (k >= n_b) return;
        src = in_b + PEDONI_HALO_HEADER_WORDS + (size_t)k * PEDONI_HALO_RECORD_WORDS;
        at = at0 + n_a + k;
    }
    pos[at] = make_float2(__uint_as_float(src[0]), __uint_as_float(src[1]));
    velx[at] = make_float4(__uint_as_float(src[2]), __uint_as_float(src[3]), 0.0f, __uint_as_float(src[4]));
    dest[at] = src[5];
}

} // namespace

struct PedoniShard {
    PedoniModel* m = nullptr;
    int32_t rank = 0, world = 1;
    std::vector<int32_t> bounds;      // world + 1 row boundaries, identical on every rank
    uint32_t cap = 0, words_each = 0; // halo list capacity (agents) / words per list
    uint32_t* d_send = nullptr;       // [down list][up list]
    uint32_t* d_recv_below = nullptr; // laid out like the band below's buffer (its UP list is used)
    uint32_t* d_recv_above = nullptr; // like the band above's buffer (its DOWN list is used)
    ncclComm_t comm = nullptr;
    bool begun = false;
    // re-cut
    uint32_t rebalance_every = 0, max_shift = 4, ticks = 0, recuts = 0;
    uint32_t bulk_cap = 0, bulk_words = 0;
    uint32_t* d_hist = nullptr;       // n_rows
    uint32_t* h_hist = nullptr;       // pinned
    uint32_t* d_bulk_send[2] = {nullptr, nullptr}; // [0] for the band below, [1] for the band above
    uint32_t* d_bulk_recv[2] = {nullptr, nullptr}; // [0] from the band below, [1] from the band above
    // overlap mode: the exchange of the NEXT tick's lists runs on its own stream while this
    // tick's interior rows are still being computed (pedoni_hip_halo_tick_begin / _end)
    bool overlap = false, in_flight = false;
    bool lists_ready = false;   // the receive buffers already hold this tick's lists
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_recv = nullptr;
    // edge-first overlapped tick: a device word the force launch sets when its edge tiles are done and
    // the communication stream's edge_wait_kernel polls (null: PEDONI_SHARD_FORM=split)
    uint32_t* d_edge_flag = nullptr;
    uint32_t* d_edge_counter = nullptr;
    uint32_t edge_seq = 0;
    uint32_t recv_seq = 0;        // d_edge_flag[2]: the lists of tick `recv_seq` are unpacked (edge_post_kernel)
    uint32_t n_edge_first = 0, n_split = 0, n_plain = 0;   // pedoni_shard_tick_forms
    // inside pedoni_shard_tick_n an edge-first tick also unpacks the NEXT tick's lists, on the communication
    // stream right behind the exchange (halo_unpack_on): the next tick then starts at its sort pass
    bool more_ticks = false;      // not the last tick of this pedoni_shard_tick_n call
    bool unpacked_ahead = false;  // the lists of the coming tick are in the model already
    // members of a local group (one process, one device) reach each other directly -- through the
    // caller's array, valid only inside pedoni_shard_local_group_tick_n
    PedoniShard** group = nullptr;
    bool local_member = false;        // has ticked as a member of a local group
    // initial bounds and slack every rank's map slice was cut with (pedoni_shard_map_rows): a
    // re-cut may move boundary b only where both adjoining bands still fit their slices
    std::vector<int32_t> bounds0;
    int32_t slice_slack = -1;          // < 0: whole maps on the device, no constraint
};

namespace {

int shard_check(PedoniShard* s)
{
    if (!s || !s->m) return fail(PEDONI_E_INVALID, "null shard");
    return bind(s->m);
}

const uint32_t* shard_below(const PedoniShard* s) { return s->rank > 0 ? s->d_recv_below : nullptr; }
const uint32_t* shard_above(const PedoniShard* s) { return s->rank + 1 < s->world ? s->d_recv_above : nullptr; }

// the per-tick neighbour exchange of the packed lists, on the model's stream (or, in overlap
// mode, on the shard's communication stream)
int shard_exchange_rccl(PedoniShard* s, hipStream_t st)
{
    if (s->world == 1) return PEDONI_OK;
    if (!s->comm) return fail(PEDONI_E_INVALID, "shard has no communicator (created without an id)");
    RcclApi& api = rccl();
    const size_t n = s->words_each;
    return rccl_group("halo exchange", [&](auto&& op) {
        if (s->rank > 0) {
            op([&] { return api.Send(s->d_send, n, ncclUint32, s->rank - 1, s->comm, st); }, "ncclSend(down list)");
            op([&] { return api.Recv(s->d_recv_below + n, n, ncclUint32, s->rank - 1, s->comm, st); }, "ncclRecv(up list of the band below)");
        }
        if (s->rank + 1 < s->world) {
            op([&] { return api.Send(s->d_send + n, n, ncclUint32, s->rank + 1, s->comm, st); }, "ncclSend(up list)");
            op([&] { return api.Recv(s->d_recv_above, n, ncclUint32, s->rank + 1, s->comm, st); }, "ncclRecv(down list of the band above)");
        }
    });
}

int shard_exchange_local(PedoniShard* s)
{
    const size_t bytes = (size_t)s->words_each * sizeof(uint32_t);
    if (s->rank > 0)
        HIP_TRY(hipMemcpyAsync(s->d_recv_below + s->words_each, s->group[s->rank - 1]->d_send + s->words_each,
                               bytes, hipMemcpyDeviceToDevice, s->m->stream));
    if (s->rank + 1 < s->world)
        HIP_TRY(hipMemcpyAsync(s->d_recv_above, s->group[s->rank + 1]->d_send, bytes, hipMemcpyDeviceToDevice,
                               s->m->stream));
    return PEDONI_OK;
}

int shard_pack(PedoniShard* s) { return halo_pack_from(s->m, s->d_send, s->cap, /*updated=*/false); }

// ---- re-cut ------------------------------------------------------------------------------------
// phase 1 (after a sort/despawn pass): this band's agents per owned row
int recut_hist(PedoniShard* s)
{
    PedoniModel* m = s->m;
    hipLaunchKernelGGL(row_hist_kernel, dim3(blocks_for((uint32_t)m->grid.rows, 256)), dim3(256), 0, m->stream,
                       m->d_cs[m->cs], m->grid.cols, m->band_lo, m->band_hi, m->grid.rows, s->d_hist);
    HIP_TRY(hipGetLastError());
    return PEDONI_OK;
}

// the cut itself: same input, same result on every rank (pedoni_shard_recut_bounds: pure host
// code, tested on the CPU)
void recut_bounds(const PedoniShard* s, const uint32_t* hist, std::vector<int32_t>& nb)
{
    nb.assign(s->bounds.size(), 0);
    pedoni_shard_recut_bounds(s->bounds.data(), s->world, hist, (uint32_t)s->m->grid.rows, s->max_shift,
                              s->bulk_cap, s->bounds0.data(), s->slice_slack, nb.data());
}

// phase 2: pack the rows this band hands over (old bounds), as decided by every rank alike
int recut_pack(PedoniShard* s, const std::vector<int32_t>& nb)
{
    PedoniModel* m = s->m;
    for (int dir = 0; dir < 2; ++dir) {
        const int32_t b = s->rank + dir;                   // boundary index: lower (dir 0) / upper (dir 1)
        int32_t row_a = 0, row_b = 0;
        if (b >= 1 && b < s->world) {
            const int32_t old = s->bounds[b], to = nb[b];
            if (dir == 1 && to < old) { row_a = to - 1; row_b = old - 1; }      // my top rows go up
            if (dir == 0 && to > old) { row_a = old + 1; row_b = to + 1; }      // my bottom rows go down
        }
        hipLaunchKernelGGL(bulk_pack_kernel, dim3(blocks_for(s->bulk_cap, 256)), dim3(256), 0, m->stream,
                           m->d_pos[m->pv], m->d_velx[m->pv], m->d_dest[m->vd], m->d_cs[m->cs],
                           m->grid.cols, row_a, row_b, s->bulk_cap, s->d_bulk_send[dir]);
    }
    HIP_TRY(hipGetLastError());
    return PEDONI_OK;
}

int recut_exchange_rccl(PedoniShard* s)
{
    RcclApi& api = rccl();
    hipStream_t st = s->m->stream;
    return rccl_group("re-cut exchange", [&](auto&& op) {
        if (s->rank > 0) {
            op([&] { return api.Send(s->d_bulk_send[0], s->bulk_words, ncclUint32, s->rank - 1, s->comm, st); }, "ncclSend(rows down)");
            op([&] { return api.Recv(s->d_bulk_recv[0], s->bulk_words, ncclUint32, s->rank - 1, s->comm, st); }, "ncclRecv(rows from below)");
        }
        if (s->rank + 1 < s->world) {
            op([&] { return api.Send(s->d_bulk_send[1], s->bulk_words, ncclUint32, s->rank + 1, s->comm, st); }, "ncclSend(rows up)");
            op([&] { return api.Recv(s->d_bulk_recv[1], s->bulk_words, ncclUint32, s->rank + 1, s->comm, st); }, "ncclRecv(rows from above)");
        }
    });
}

int recut_exchange_local(PedoniShard* s)
{
    const size_t bytes = (size_t)s->bulk_words * sizeof(uint32_t);
    if (s->rank > 0)
        HIP_TRY(hipMemcpyAsync(s->d_bulk_recv[0], s->group[s->rank - 1]->d_bulk_send[1], bytes,
                               hipMemcpyDeviceToDevice, s->m->stream));
    if (s->rank + 1 < s->world)
        HIP_TRY(hipMemcpyAsync(s->d_bulk_recv[1], s->group[s->rank + 1]->d_bulk_send[0], bytes,
                               hipMemcpyDeviceToDevice, s->m->stream));
    return PEDONI_OK;
}

// phase 3: take the incoming rows, adopt the new band, sort again (sfm.rs:58-77 semantics: a
// second pass over unchanged positions drops nobody and only re-orders)
int recut_apply(PedoniShard* s, const std::vector<int32_t>& nb)
{
    PedoniModel* m = s->m;
    const bool moved = nb[s->rank] != s->bounds[s->rank] || nb[s->rank + 1] != s->bounds[s->rank + 1];
    s->bounds = nb;
    if (!moved) return PEDONI_OK;
    // outer bands have no list from outside: an all-zero header stands in
    if (s->rank == 0) HIP_TRY(hipMemsetAsync(s->d_bulk_recv[0], 0, 4 * sizeof(uint32_t), m->stream));
    if (s->rank + 1 == s->world) HIP_TRY(hipMemsetAsync(s->d_bulk_recv[1], 0, 4 * sizeof(uint32_t), m->stream));
    TRY(ensure_capacity(m, m->n_upper + 2 * s->bulk_cap));
    hipLaunchKernelGGL(bulk_unpack_kernel, dim3(blocks_for(2 * s->bulk_cap, 256)), dim3(256), 0, m->stream,
                       s->d_bulk_recv[0], s->d_bulk_recv[1], s->bulk_cap, m->n_upper, m->d_pos[m->pv],
                       m->d_velx[m->pv], m->d_dest[m->vd], m->d_halo);
    HIP_TRY(hipGetLastError());
    m->gap_end = m->n_upper;
    m->n_upper += 2 * s->bulk_cap;
    m->band_lo = nb[s->rank];
    m->band_hi = nb[s->rank + 1];
    m->keys_valid = false;          // every stored agent is keyed afresh against the new band
    m->halo_keys_done = false;
    m->sorted = false;
    m->drop_graphs();
    TRY(sort_despawn(m));
    s->recuts += 1;
    return PEDONI_OK;
}

bool recut_due(const PedoniShard* s)
{
    return s->rebalance_every && s->world > 1 && (s->ticks % s->rebalance_every) == s->rebalance_every - 1;
}

// one tick of one rank over RCCL
// with profiling on, only every profile_every-th tick is event-timed (as in pedoni_hip_tick_n)
struct SampledProfile {
    PedoniModel* m;
    explicit SampledProfile(PedoniModel* m_) : m(m_)
    {
        m->profile_now = m->sampled(m->tick_counter);
        m->tick_counter += 1;
    }
    ~SampledProfile() { m->profile_now = true; }
};

// the lists of THIS tick: already on their way (overlap mode) or exchanged now
int shard_get_lists(PedoniShard* s)
{
    if (s->in_flight) {
        if (s->unpacked_ahead) {
            // the lists are being unpacked on the communication stream: the sort pass's first launch
            // waits for them itself (scan_rows_kernel), no event between the force launch and it
            s->m->scan_wait_flag = s->d_edge_flag + 2;
            s->m->scan_wait_seq = s->recv_seq;
        } else {
            HIP_TRY(hipStreamWaitEvent(s->m->stream, s->ev_recv, 0));
        }
        s->in_flight = false;
        return PEDONI_OK;
    }
    if (s->lists_ready) {
        s->lists_ready = false;
        return PEDONI_OK;
    }
    return shard_exchange_rccl(s, s->m->stream);
}

// the received lists into the model -- unless the tick before has done that already
int shard_unpack(PedoniShard* s)
{
    if (s->unpacked_ahead) { s->unpacked_ahead = false; return PEDONI_OK; }
    return pedoni_hip_halo_unpack(s->m, shard_below(s), shard_above(s), s->cap);
}

// overlap mode: the freshly packed lists leave on the communication stream
int shard_start_next(PedoniShard* s)
{
    if (!s->overlap) return PEDONI_OK;
    HIP_TRY(hipEventRecord(s->ev_packed, s->m->stream));
    HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->ev_packed, 0));
    TRY(shard_exchange_rccl(s, s->comm_stream));
    HIP_TRY(hipEventRecord(s->ev_recv, s->comm_stream));
    s->in_flight = true;
    return PEDONI_OK;
}

// Overlapped tick: the pack of the freshly updated edge rows and the exchange of the NEXT tick's lists
// run on the shard's own high-priority stream under the rest of this tick's force launch; the next tick
// joins on one event that has long fired.  Two forms.
//
// EDGE-FIRST (bands large enough for the one-lane-per-agent kernel; the form bench.py's 1e6-agent bands
// run): ONE force launch over the band -- the plain tick's -- whose first workgroups take the edge rows'
// tiles and set a device word when the last of them has released its records
// (force_kernel_queue_edge_first); the communication stream holds a one-wave kernel that looks at that
// word every ~2 us (edge_wait_kernel; bounded: STATUS_EDGE_WAIT after 5 s):
//     model stream:  unpack, sort/despawn, force(edge tiles first ... interior tiles) ........  [join]
//     comm stream :                          [wave polls word >= seq] pack, ncclSend/Recv, [record]
//                                                    ... [unpack of the NEXT tick's lists] [record]
// The model's stream carries the plain tick's launches minus the pack -- and, inside pedoni_shard_tick_n,
// minus the unpack: the lists of the coming tick are unpacked on the communication stream right behind
// the exchange that brought them (halo_unpack_on; they land outside the agents the running force launch
// works on), so the next tick starts at its sort pass -- whose first launch looks at a word the
// communication stream stores behind that unpack (edge_post_kernel, scan_rows_kernel's wait_flag) instead
// of the model's stream waiting on an event: a cross-stream event wait between the force launch and the
// scan was ~5 us of idle device.  The kernel timeline of the tick is then scan, place, reorder, force
// back to back (113 us traced; 128 with unpack and event wait on the model's stream, 131 plain,
// profiles/r03_shard_timeline.txt): what is left over the unsharded tick is the reorder launch.  The pack
// is done 28 us into the 86-us force launch.
//
// SPLIT (small bands, whose force kernel is the 2-4-lanes-per-agent one; PEDONI_SHARD_FORM=split): the
// few rows beside the band's edges FIRST (a small launch), then the interior rows (the bulk):
//     model stream:  unpack, sort/despawn, force(edge rows) [fork] force(interior rows) ......  [join]
//     comm stream :                                  [wait fork] pack, ncclSend/Recv, [record]
// What it costs over the plain tick is the edge launch (8 000 agents: 9 us on the 4-lanes-per-agent
// kernel, 15 us on the one-lane kernel) and ~7 us of event record between the two force launches:
// 140-144 us per tick at 1e6 agents.
//
// bench.py times the plain and the overlapped tick on the node and keeps the faster.  Forms measured and
// dropped on the way (kernel timelines, profiles/r03_shard_timeline.txt; one GPU, nothing on the wire,
// plain tick 124-125 us): round 2's (edge rows + pack on the model's stream, interior on a side stream,
// exchange on a third: three cross-stream hops of 7-20 us each on the critical path, and a 1024-thread
// pack workgroup that waited 50 us for 16 free wave slots on one CU) 152 us; edge rows + pack + exchange
// on a high-priority stream BESIDE the interior launch: 137 us, but the interior's workgroups took the
// chip first and the edge launch, 33 workgroups, trickled in over 73 us -- the pack ended WITH the
// interior, the exchange hidden under nothing; the edge-first launch released through
// hipStreamWaitValue32: 152 us (the runtime's wait is a kernel that polls without pause: place kernel
// 26 us instead of 18.5, force kernel 103 instead of 92 while it is in flight).
int shard_tick_split(PedoniShard* s)
{
    PedoniModel* m = s->m;
    TRY(shard_unpack(s));
    TRY(sort_despawn(m));
    if (m->band_hi - m->band_lo < 6 || m->force_simple) {        // band too thin to split: the plain sequence
        TRY(update_states(m));
        TRY(shard_pack(s));
        s->n_plain += 1;
        return shard_start_next(s);
    }
    m->edge_flag = s->d_edge_flag;
    m->edge_counter = s->d_edge_counter;
    if (edge_first_ready(m)) {
        m->edge_seq = ++s->edge_seq;
        TRY(launch_force(m, nullptr, /*part=*/3));
        // (only now that the launch which WILL store the word is in the queue; the waiting wave gives up
        // after 5 s of the 100 MHz clock and raises STATUS_EDGE_WAIT)
        hipLaunchKernelGGL(edge_wait_kernel, dim3(1), dim3(64), 0, s->comm_stream, s->d_edge_flag, s->edge_seq,
                           m->d_live + 1, 500000000ull);
        HIP_TRY(hipGetLastError());
        TRY(halo_pack_from(m, s->d_send, s->cap, /*updated=*/true, s->comm_stream));
        TRY(shard_exchange_rccl(s, s->comm_stream));
        after_update(m);
        if (s->more_ticks && !halo_unpack_would_tighten(m, s->cap, shard_below(s) != nullptr, shard_above(s) != nullptr)) {
            // the coming tick's unpack, here and now: off the model's stream, under this tick's force launch
            // (not the one unpack in 8 that re-reads the live count: see halo_unpack_would_tighten)
            TRY(halo_unpack_on(m, shard_below(s), shard_above(s), s->cap, s->comm_stream));
            s->unpacked_ahead = true;
            hipLaunchKernelGGL(edge_post_kernel, dim3(1), dim3(64), 0, s->comm_stream, s->d_edge_flag + 2, ++s->recv_seq);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(s->ev_recv, s->comm_stream));
        s->in_flight = true;
        s->n_edge_first += 1;
        return PEDONI_OK;
    }
    TRY(launch_force(m, nullptr, /*part=*/1));                    // edge rows (ghost rows are only NaN-marked)
    HIP_TRY(hipEventRecord(s->ev_packed, m->stream));             // fork: the edge rows are updated
    // (the interior launch is SUBMITTED before the comm stream's work: the RCCL calls cost the host
    // ~10 us, which the device would otherwise spend idle between the two force launches)
    TRY(launch_force(m, nullptr, /*part=*/2));                    // interior rows
    HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->ev_packed, 0));
    TRY(halo_pack_from(m, s->d_send, s->cap, /*updated=*/true, s->comm_stream));
    TRY(shard_exchange_rccl(s, s->comm_stream));
    HIP_TRY(hipEventRecord(s->ev_recv, s->comm_stream));
    s->in_flight = true;                                          // the next tick joins on ev_recv
    after_update(m);
    s->n_split += 1;
    return PEDONI_OK;
}

// A tick that failed midway leaves the shard's hand-offs (lists in flight, lists unpacked ahead, the band's
// arrays half updated) in no state a further tick could continue from: settle what is under way and refuse
// further ticks until the caller reloads the band and calls pedoni_shard_begin again.
void shard_poison(PedoniShard* s)
{
    if (s->comm_stream) hipStreamSynchronize(s->comm_stream);
    if (s->m && s->m->stream) hipStreamSynchronize(s->m->stream);
    s->in_flight = s->unpacked_ahead = s->more_ticks = s->lists_ready = false;
    if (s->m) s->m->scan_wait_flag = nullptr;
    s->begun = false;
}

int shard_tick_rccl_body(PedoniShard* s);
int shard_tick_rccl(PedoniShard* s)
{
    // (keep the failing call's message: the settling below makes HIP calls of its own)
    const int rc = shard_tick_rccl_body(s);
    if (rc != PEDONI_OK) {
        const std::string why = g_last_error;
        shard_poison(s);
        g_last_error = why + " (the shard is stopped: reload the band and call pedoni_shard_begin)";
    }
    return rc;
}

int shard_tick_rccl_body(PedoniShard* s)
{
    PedoniModel* m = s->m;
    SampledProfile sampled(m);
    TRY(shard_get_lists(s));
    if (!recut_due(s)) {
        if (s->overlap) {
            TRY(shard_tick_split(s));
        } else {
            TRY(shard_unpack(s));                          // (= pedoni_hip_halo_tick)
            TRY(sort_despawn(m));
            TRY(update_states(m));
            TRY(shard_pack(s));
            s->n_plain += 1;
        }
    } else {
        s->n_plain += 1;
        TRY(shard_unpack(s));
        TRY(sort_despawn(m));
        TRY(recut_hist(s));
        NCCL_TRY(rccl().AllReduce(s->d_hist, s->d_hist, (size_t)m->grid.rows, ncclUint32, ncclSum, s->comm, m->stream));
        HIP_TRY(hipMemcpyAsync(s->h_hist, s->d_hist, (size_t)m->grid.rows * sizeof(uint32_t), hipMemcpyDeviceToHost,
                               m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));          // stop the world: once per re-cut
        std::vector<int32_t> nb;
        recut_bounds(s, s->h_hist, nb);
        TRY(recut_pack(s, nb));
        TRY(recut_exchange_rccl(s));
        TRY(recut_apply(s, nb));
        TRY(update_states(m));
        TRY(shard_pack(s));
        TRY(shard_start_next(s));
    }
    s->ticks += 1;
    return PEDONI_OK;
}

} // namespace

void shard_detach_model(PedoniModel* m)
{
    if (PedoniShard* s = m->shard) {
        // the communication stream may still hold the last overlapped tick's edge wait, pack and send, which
        // read the model's arrays: drained HERE, before the caller frees them (not left to hipFree's implicit
        // device synchronisation)
        if (s->comm_stream) hipStreamSynchronize(s->comm_stream);
        s->in_flight = s->unpacked_ahead = s->more_ticks = false;
        s->begun = false;
        s->m = nullptr;
    }
    m->shard = nullptr;
    m->edge_flag = m->edge_counter = nullptr;
    m->scan_wait_flag = nullptr;
}

extern "C" {

int pedoni_shard_unique_id(uint8_t id[PEDONI_SHARD_ID_BYTES])
{
    if (!id) return fail(PEDONI_E_INVALID, "null id");
    RcclApi& api = rccl();
    if (!api.handle || !api.error.empty()) return fail(PEDONI_E_HIP, api.error);
    static_assert(sizeof(ncclUniqueId) == PEDONI_SHARD_ID_BYTES, "RCCL id size");
    ncclUniqueId u;
    NCCL_TRY(api.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return PEDONI_OK;
}

int pedoni_shard_map_rows(int32_t row_begin, int32_t row_end, int32_t slack_rows, float neighbor_grid_unit,
                          float field_unit, uint32_t field_rows, uint32_t* map_row_begin, uint32_t* map_row_end)
{
    if (!map_row_begin || !map_row_end || row_begin >= row_end || slack_rows < 0 || !(neighbor_grid_unit > 0.0f) ||
        !(field_unit > 0.0f) || field_rows == 0)
        return fail(PEDONI_E_INVALID, "shard_map_rows: bad arguments");
    // agents of the band sit in grid rows row_begin-1 .. row_end (ghost rows included) and step at
    // most one more row per tick; every sample point p / unit - 0.5 is the centre of a 4 x 4 patch
    const double y0 = (double)(row_begin - 2 - slack_rows) * neighbor_grid_unit;
    const double y1 = (double)(row_end + 2 + slack_rows) * neighbor_grid_unit;
    const double t0 = std::floor(y0 / field_unit - 0.5) - 3.0, t1 = std::ceil(y1 / field_unit - 0.5) + 4.0;
    *map_row_begin = (uint32_t)std::max(0.0, std::min(t0, (double)field_rows - 1.0));
    *map_row_end = (uint32_t)std::max((double)*map_row_begin + 1.0, std::min(t1, (double)field_rows));
    return PEDONI_OK;
}

int pedoni_shard_recut_bounds(const int32_t* bounds, int32_t world, const uint32_t* row_counts, uint32_t n_rows,
                              uint32_t max_shift, uint32_t bulk_cap, const int32_t* bounds0, int32_t map_slack_rows,
                              int32_t* bounds_out)
{
    const int32_t min_rows = 6;
    if (!bounds || !row_counts || !bounds_out || world < 1 || (map_slack_rows >= 0 && !bounds0))
        return fail(PEDONI_E_INVALID, "recut_bounds: bad arguments");
    for (int32_t b = 0; b <= world; ++b) bounds_out[b] = bounds[b];
    if ((uint64_t)world * (uint64_t)min_rows > n_rows) return PEDONI_OK;      // too few rows to move anything
    std::vector<int32_t> ideal((size_t)world + 1);
    TRY(pedoni_shard_balanced_bounds(row_counts, n_rows, world, min_rows, ideal.data()));
    for (int32_t b = 1; b < world; ++b) {
        const int32_t old = bounds[b];
        int32_t want = std::max(old - (int32_t)max_shift, std::min(old + (int32_t)max_shift, ideal[b]));
        // the donor hands over rows it held BEFORE the cut; every band keeps >= min_rows
        want = std::max(want, std::max(bounds[b - 1] + min_rows, bounds_out[b - 1] + min_rows));
        want = std::min(want, bounds[b + 1] - min_rows);
        if (want < bounds_out[b - 1] + min_rows) want = old;                 // cannot satisfy both: leave it
        // bands that hold a slice of the field maps stay inside it: every rank cut its slice from
        // the INITIAL bounds with the same slack, so every rank evaluates the same predicate
        auto fits = [&](int32_t to) {
            return map_slack_rows < 0 || (to <= bounds0[b] + map_slack_rows && to >= bounds0[b] - map_slack_rows);
        };
        while (want != old && !fits(want)) want += want < old ? 1 : -1;
        // the rows handed over must fit one bulk list: the donor sends rows [to-1, old-1) when the
        // cut moves down, [old+1, to+1) when it moves up
        auto moved = [&](int32_t to) {
            uint64_t n = 0;
            const int32_t a = to < old ? to - 1 : old + 1, e = to < old ? old - 1 : to + 1;
            for (int32_t r = std::max(a, 0); r < std::min(e, (int32_t)n_rows); ++r) n += row_counts[r];
            return n;
        };
        while (want != old && moved(want) > bulk_cap) want += want < old ? 1 : -1;
        bounds_out[b] = want;
    }
    return PEDONI_OK;
}

int pedoni_shard_balanced_bounds(const uint32_t* row_counts, uint32_t n_rows, int32_t world, int32_t min_rows,
                                 int32_t* bounds_out)
{
    if (!row_counts || !bounds_out || world < 1 || min_rows < 1 || (uint64_t)world * (uint64_t)min_rows > n_rows)
        return fail(PEDONI_E_INVALID, "balanced_bounds: bad arguments (need world * min_rows <= n_rows)");
    uint64_t total = 0;
    for (uint32_t r = 0; r < n_rows; ++r) total += row_counts[r];
    bounds_out[0] = 0;
    bounds_out[world] = (int32_t)n_rows;
    uint64_t run = 0;
    uint32_t r = 0;
    for (int32_t b = 1; b < world; ++b) {
        // first row boundary at which the agents below reach b / world of the crowd
        const uint64_t target = (total * (uint64_t)b + (uint64_t)world / 2) / (uint64_t)world;
        while (r < n_rows && run + row_counts[r] <= target) run += row_counts[r++];
        int32_t cut = (int32_t)r;
        cut = std::max(cut, bounds_out[b - 1] + min_rows);
        cut = std::min(cut, (int32_t)n_rows - (world - b) * min_rows);
        bounds_out[b] = cut;
        while ((int32_t)r < cut) run += row_counts[r++];   // keep the running sum at the cut
    }
    return PEDONI_OK;
}

int pedoni_shard_create(PedoniModel* m, int32_t rank, int32_t world, const uint8_t* id, const int32_t* row_bounds,
                        uint32_t halo_cap, PedoniShard** out)
{
    TRY(bind(m));
    if (!out || !row_bounds || world < 1 || rank < 0 || rank >= world || halo_cap == 0)
        return fail(PEDONI_E_INVALID, "shard_create: bad arguments");
    for (int32_t r = 0; r < world; ++r)
        if (row_bounds[r + 1] - row_bounds[r] < 2 || row_bounds[0] != 0 || row_bounds[world] != m->grid.rows)
            return fail(PEDONI_E_INVALID, "shard_create: row_bounds must rise from 0 to the grid's rows, >= 2 rows per band");
    if (m->shard) return fail(PEDONI_E_INVALID, "shard_create: the model already belongs to a shard");
    PedoniShard* s = new PedoniShard();
    s->m = m;
    m->shard = s;      // pedoni_hip_destroy(m) before pedoni_shard_destroy(s) detaches instead of dangling
    s->rank = rank;
    s->world = world;
    s->bounds.assign(row_bounds, row_bounds + world + 1);
    s->bounds0 = s->bounds;
    s->cap = halo_cap;
    s->words_each = PEDONI_HALO_HEADER_WORDS + halo_cap * PEDONI_HALO_RECORD_WORDS;
    auto bail = [&](int rc) { pedoni_shard_destroy(s); return rc; };
    int rc = pedoni_hip_set_band(m, row_bounds[rank], row_bounds[rank + 1], halo_cap);
    if (rc) return bail(rc);
    const size_t words = 2 * (size_t)s->words_each;
    for (uint32_t** p : {&s->d_send, &s->d_recv_below, &s->d_recv_above}) {
        if ((rc = dev_alloc(p, words)) != PEDONI_OK) return bail(rc);
        if (hipMemset(*p, 0, words * sizeof(uint32_t)) != hipSuccess) return bail(fail(PEDONI_E_HIP, "hipMemset failed"));
    }
    if (id) {
        RcclApi& api = rccl();
        if (!api.handle || !api.error.empty()) return bail(fail(PEDONI_E_HIP, api.error));
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        ncclResult_t r = api.CommInitRank(&s->comm, world, u, rank);
        if (r != ncclSuccess) {
            s->comm = nullptr;
            return bail(fail(PEDONI_E_HIP, std::string("ncclCommInitRank: ") + api.GetErrorString(r)));
        }
    }
    *out = s;
    return PEDONI_OK;
}

void pedoni_shard_destroy(PedoniShard* s)
{
    if (!s) return;
    if (s->m) {
        hipSetDevice(s->m->device);
        if (s->m->stream) hipStreamSynchronize(s->m->stream);
        s->m->shard = nullptr;
    }
    if (s->comm_stream) { hipStreamSynchronize(s->comm_stream); hipStreamDestroy(s->comm_stream); }
    if (s->m) s->m->edge_flag = s->m->edge_counter = nullptr;
    hipFree(s->d_edge_flag);
    if (s->ev_packed) hipEventDestroy(s->ev_packed);
    if (s->ev_recv) hipEventDestroy(s->ev_recv);
    if (s->comm && rccl().CommDestroy) rccl().CommDestroy(s->comm);
    hipFree(s->d_send); hipFree(s->d_recv_below); hipFree(s->d_recv_above); hipFree(s->d_hist);
    for (int k = 0; k < 2; ++k) { hipFree(s->d_bulk_send[k]); hipFree(s->d_bulk_recv[k]); }
    if (s->h_hist) hipHostFree(s->h_hist);
    delete s;
}

int pedoni_shard_begin(PedoniShard* s)
{
    TRY(shard_check(s));
    TRY(sort_despawn(s->m));
    TRY(shard_pack(s));
    s->begun = true;
    return PEDONI_OK;
}

int pedoni_shard_tick_n(PedoniShard* s, uint32_t steps)
{
    TRY(shard_check(s));
    if (!s->begun) return fail(PEDONI_E_INVALID, "shard_tick_n: call pedoni_shard_begin after loading the band");
    if (s->local_member) return fail(PEDONI_E_INVALID, "shard_tick_n: a member of a local group ticks with pedoni_shard_local_group_tick_n");
    for (uint32_t k = 0; k < steps; ++k) {
        s->more_ticks = k + 1 < steps;
        const int rc = shard_tick_rccl(s);
        s->more_ticks = false;
        if (rc != PEDONI_OK) return rc;
    }
    return PEDONI_OK;
}

int pedoni_shard_owned_count(PedoniShard* s, int32_t* count)
{
    TRY(shard_check(s));
    return pedoni_hip_owned_count(s->m, count);
}

int pedoni_shard_band(PedoniShard* s, int32_t* row_begin, int32_t* row_end)
{
    if (!s) return fail(PEDONI_E_INVALID, "null shard");
    if (row_begin) *row_begin = s->bounds[s->rank];
    if (row_end) *row_end = s->bounds[s->rank + 1];
    return PEDONI_OK;
}

int pedoni_shard_set_rebalance(PedoniShard* s, uint32_t every_ticks, uint32_t max_rows_per_step,
                               int32_t map_slack_rows)
{
    TRY(shard_check(s));
    PedoniModel* m = s->m;
    if (every_ticks && (max_rows_per_step == 0 || max_rows_per_step > 64))
        return fail(PEDONI_E_INVALID, "set_rebalance: max_rows_per_step must be 1 .. 64");
    const bool whole_maps = m->field.y_lo == 0 && m->field.y_hi == m->field.rows;
    if (every_ticks && map_slack_rows < 0 && !whole_maps)
        return fail(PEDONI_E_INVALID, "set_rebalance: this model holds a slice of the field maps; state the slack it was cut with");
    if (every_ticks && map_slack_rows >= 0) {
        uint32_t a = 0, b = 0;
        TRY(pedoni_shard_map_rows(s->bounds0[s->rank], s->bounds0[s->rank + 1], map_slack_rows, m->grid.unit,
                                  m->field.unit, (uint32_t)m->field.rows, &a, &b));
        if ((int32_t)a < m->field.y_lo || (int32_t)b > m->field.y_hi)
            return fail(PEDONI_E_INVALID, "set_rebalance: the model's field-map rows do not cover its band plus that slack");
    }
    s->slice_slack = map_slack_rows;
    s->rebalance_every = every_ticks;
    if (!every_ticks) return PEDONI_OK;
    s->max_shift = max_rows_per_step;
    if (!s->d_hist) {
        TRY(dev_alloc(&s->d_hist, (size_t)m->grid.rows));
        HIP_TRY(hipHostMalloc((void**)&s->h_hist, (size_t)m->grid.rows * sizeof(uint32_t), hipHostMallocDefault));
    }
    // one bulk list holds max_shift rows at 1.5x the halo capacity's row estimate (a halo list
    // is sized for about two rows); the cut is limited to what fits, so this is never exceeded
    const uint32_t bulk_cap = std::max(s->cap, s->cap * s->max_shift);
    if (bulk_cap != s->bulk_cap) {
        for (int k = 0; k < 2; ++k) {
            hipFree(s->d_bulk_send[k]); hipFree(s->d_bulk_recv[k]);
            s->d_bulk_send[k] = s->d_bulk_recv[k] = nullptr;
        }
        s->bulk_cap = bulk_cap;
        s->bulk_words = PEDONI_HALO_HEADER_WORDS + bulk_cap * PEDONI_HALO_RECORD_WORDS;
        for (int k = 0; k < 2; ++k) {
            TRY(dev_alloc(&s->d_bulk_send[k], (size_t)s->bulk_words));
            TRY(dev_alloc(&s->d_bulk_recv[k], (size_t)s->bulk_words));
            HIP_TRY(hipMemset(s->d_bulk_recv[k], 0, (size_t)s->bulk_words * sizeof(uint32_t)));
        }
    }
    return PEDONI_OK;
}

int pedoni_shard_set_overlap(PedoniShard* s, int32_t on)
{
    TRY(shard_check(s));
    if (s->local_member) return fail(PEDONI_E_INVALID, "set_overlap: not for members of a local group");
    if (s->in_flight) {                                   // settle the exchange that is under way
        HIP_TRY(hipStreamWaitEvent(s->m->stream, s->ev_recv, 0));
        HIP_TRY(hipStreamSynchronize(s->m->stream));
        // its lists are in the receive buffers: the next tick must not exchange again
        s->in_flight = false;
        s->lists_ready = true;
    }
    if (on && !s->comm_stream) {
        // (highest priority: its small launches take the wave slots the interior kernel frees)
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&s->comm_stream, hipStreamNonBlocking, greatest));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_packed, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_recv, hipEventDisableTiming));
        // the edge-first form's signal word (PEDONI_SHARD_FORM=split keeps the two-launch form)
        const char* form = std::getenv("PEDONI_SHARD_FORM");
        if (!(form && std::string(form) == "split")) {
            HIP_TRY(hipMalloc((void**)&s->d_edge_flag, 3 * sizeof(uint32_t)));     // flag, arrival counter, unpacked word
            s->d_edge_counter = s->d_edge_flag + 1;
            HIP_TRY(hipMemsetAsync(s->d_edge_flag, 0, 3 * sizeof(uint32_t), s->comm_stream));
            HIP_TRY(hipStreamSynchronize(s->comm_stream));
        }
    }
    s->overlap = on != 0;
    return PEDONI_OK;
}

int pedoni_shard_tick_forms(PedoniShard* s, uint32_t* edge_first, uint32_t* split, uint32_t* plain)
{
    TRY(shard_check(s));
    if (edge_first) *edge_first = s->n_edge_first;
    if (split) *split = s->n_split;
    if (plain) *plain = s->n_plain;
    return PEDONI_OK;
}

int pedoni_shard_selftest(PedoniShard* s)
{
    TRY(shard_check(s));
    if (!s->comm) return s->world == 1 ? PEDONI_OK : fail(PEDONI_E_INVALID, "shard_selftest: no communicator");
    if (s->world == 1) {
        // a lone rank still proves the dlopen'ed ncclSend / ncclRecv work: one token to itself
        RcclApi& api = rccl();
        const uint32_t token = 0x5E1F0001u;
        uint32_t got = 0;
        HIP_TRY(hipMemcpyAsync(s->d_send, &token, sizeof token, hipMemcpyHostToDevice, s->m->stream));
        TRY(rccl_group("self-addressed token", [&](auto&& op) {
            op([&] { return api.Send(s->d_send, 1, ncclUint32, 0, s->comm, s->m->stream); }, "ncclSend");
            op([&] { return api.Recv(s->d_recv_above, 1, ncclUint32, 0, s->comm, s->m->stream); }, "ncclRecv");
        }));
        HIP_TRY(hipMemcpyAsync(&got, s->d_recv_above, sizeof got, hipMemcpyDeviceToHost, s->m->stream));
        HIP_TRY(hipStreamSynchronize(s->m->stream));
        HIP_TRY(hipMemsetAsync(s->d_send, 0, sizeof token, s->m->stream));
        HIP_TRY(hipMemsetAsync(s->d_recv_above, 0, sizeof token, s->m->stream));
        HIP_TRY(hipStreamSynchronize(s->m->stream));
        return got == token ? PEDONI_OK : fail(PEDONI_E_HIP, "shard_selftest: the self-addressed token did not arrive");
    }
    // tokens travel through the very buffers / calls of the per-tick exchange; the lists are
    // re-packed by pedoni_shard_begin afterwards
    PedoniModel* m = s->m;
    const uint32_t n = s->words_each;
    std::vector<uint32_t> tok(2 * (size_t)n, 0u);
    tok[0] = 0xD0000000u | (uint32_t)s->rank;              // first word of my DOWN list
    tok[n] = 0xA0000000u | (uint32_t)s->rank;              // first word of my UP list
    HIP_TRY(hipMemcpyAsync(s->d_send, tok.data(), tok.size() * sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    TRY(shard_exchange_rccl(s, m->stream));
    uint32_t got_below = 0, got_above = 0;
    if (s->rank > 0)
        HIP_TRY(hipMemcpyAsync(&got_below, s->d_recv_below + n, sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream));
    if (s->rank + 1 < s->world)
        HIP_TRY(hipMemcpyAsync(&got_above, s->d_recv_above, sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemsetAsync(s->d_send, 0, tok.size() * sizeof(uint32_t), m->stream));
    HIP_TRY(hipMemsetAsync(s->d_recv_below, 0, tok.size() * sizeof(uint32_t), m->stream));
    HIP_TRY(hipMemsetAsync(s->d_recv_above, 0, tok.size() * sizeof(uint32_t), m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (s->rank > 0 && got_below != (0xA0000000u | (uint32_t)(s->rank - 1)))
        return fail(PEDONI_E_HIP, "shard_selftest: wrong token from the band below");
    if (s->rank + 1 < s->world && got_above != (0xD0000000u | (uint32_t)(s->rank + 1)))
        return fail(PEDONI_E_HIP, "shard_selftest: wrong token from the band above");
    return PEDONI_OK;
}

int pedoni_shard_local_group_tick_n(PedoniShard** shards, uint32_t n_shards, uint32_t steps)
{
    if (!shards || n_shards == 0) return fail(PEDONI_E_INVALID, "local group: no shards");
    for (uint32_t r = 0; r < n_shards; ++r) {
        PedoniShard* s = shards[r];
        if (!s || s->rank != (int32_t)r || s->world != (int32_t)n_shards || s->comm || !s->begun ||
            s->m->stream != shards[0]->m->stream || s->bounds != shards[0]->bounds ||
            s->rebalance_every != shards[0]->rebalance_every || s->bulk_cap != shards[0]->bulk_cap)
            return fail(PEDONI_E_INVALID, "local group: shards must be ranks 0..n-1 of one world, begun, without a "
                                          "communicator, on ONE stream, with equal bounds and re-cut settings");
    }
    // the members reach each other through the caller's array only while this call runs (the
    // array may be a temporary of the caller's); `local_member` keeps them off pedoni_shard_tick_n
    struct Membership {
        PedoniShard** v; uint32_t n;
        Membership(PedoniShard** v_, uint32_t n_) : v(v_), n(n_) { for (uint32_t r = 0; r < n; ++r) { v[r]->group = v; v[r]->local_member = true; } }
        ~Membership() { for (uint32_t r = 0; r < n; ++r) v[r]->group = nullptr; }
    } membership(shards, n_shards);
    for (uint32_t k = 0; k < steps; ++k) {
        for (uint32_t r = 0; r < n_shards; ++r) { TRY(shard_check(shards[r])); TRY(shard_exchange_local(shards[r])); }
        if (!recut_due(shards[0])) {
            for (uint32_t r = 0; r < n_shards; ++r) {
                PedoniShard* s = shards[r];
                TRY(pedoni_hip_halo_tick(s->m, shard_below(s), shard_above(s), s->d_send, s->cap));
            }
        } else {
            const int32_t n_rows = shards[0]->m->grid.rows;
            std::vector<uint32_t> hist((size_t)n_rows, 0u);
            for (uint32_t r = 0; r < n_shards; ++r) {
                PedoniShard* s = shards[r];
                TRY(pedoni_hip_halo_unpack(s->m, shard_below(s), shard_above(s), s->cap));
                TRY(sort_despawn(s->m));
                TRY(recut_hist(s));
                HIP_TRY(hipMemcpyAsync(s->h_hist, s->d_hist, (size_t)n_rows * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                       s->m->stream));
            }
            HIP_TRY(hipStreamSynchronize(shards[0]->m->stream));
            for (uint32_t r = 0; r < n_shards; ++r)        // what ncclAllReduce(sum) does across processes
                for (int32_t i = 0; i < n_rows; ++i) hist[(size_t)i] += shards[r]->h_hist[i];
            std::vector<int32_t> nb;
            recut_bounds(shards[0], hist.data(), nb);
            for (uint32_t r = 0; r < n_shards; ++r) TRY(recut_pack(shards[r], nb));
            for (uint32_t r = 0; r < n_shards; ++r) TRY(recut_exchange_local(shards[r]));
            for (uint32_t r = 0; r < n_shards; ++r) {
                PedoniShard* s = shards[r];
                TRY(recut_apply(s, nb));
                TRY(update_states(s->m));
                TRY(shard_pack(s));
            }
        }
        for (uint32_t r = 0; r < n_shards; ++r) shards[r]->ticks += 1;
    }
    return PEDONI_OK;
}

} // extern "C"
