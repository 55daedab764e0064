// pedoni_hip.hip -- C ABI of the MI355X social-force backend (see include/pedoni_hip.h).
//
// Host orchestration of the device-resident tick.  One PedoniModel = one GPU's agents.
// Nothing here falls back to a CPU implementation: without a HIP device every entry point
// that would compute fails with PEDONI_E_NO_DEVICE / PEDONI_E_HIP.
#include "pedoni_hip.h"
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace pedoni;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(PEDONI_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

#define TRY(expr)                                                                            \
    do {                                                                                     \
        int rc_ = (expr);                                                                    \
        if (rc_ != PEDONI_OK) return rc_;                                                    \
    } while (0)

constexpr size_t TRACE_WAVES = 1u << 18;   // PEDONI_FORCE_TRACE: one 64-byte record per wave, up to 16.7 M agents

const char* const KERNEL_NAMES[PEDONI_N_KERNELS] = {
    "bin", "scan", "slot", "reorder", "force_integrate", "halo_pack", "halo_unpack", "other",
};

// ---- build-owned RNG (same specification as oracle/oracle_util.c) ----------------------
// The reference draws desired speeds from the unseeded global fastrand generator
// (sfm.rs:54), so there is no reference stream to follow: WyRand step, 24-bit f32,
// Irwin-Hall(12) normal approximation.
struct Rng {
    uint64_t s;
    uint64_t next()
    {
        s += 0xa0761d6478bd642fULL;
        __uint128_t t = (__uint128_t)s * (__uint128_t)(s ^ 0xe7037ed1a0b428dbULL);
        return (uint64_t)(t >> 64) ^ (uint64_t)t;
    }
    float f32() { return (float)(next() >> 40) * 0x1.0p-24f; }
    float normal_approx(float mu, float sigma)
    {
        float acc = 0.0f;
        for (int i = 0; i < 12; ++i) acc += f32();
        return mu + sigma * (acc - 6.0f);
    }
};

struct EventPair {
    hipEvent_t a, b;
    int kernel;
};

// ---- roctx ranges (the reference brackets spawn / update with Instant::now, lib.rs:68-91) ---
// PEDONI_ROCTX=1: every pass and every kernel launch of a tick is bracketed by a named roctx range, so
// a `rocprofv3 --marker-trace --kernel-trace` timeline shows the tick's structure (tools/roctx_trace.sh).
// The marker library is resolved with dlopen on first use and only then: without the switch (the
// default) not one call is made and nothing is loaded.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx& roctx()
{
    static const Roctx api = [] {
        Roctx r;
        const char* on = std::getenv("PEDONI_ROCTX");
        if (!on || on[0] != '1') return r;
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                r.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                r.pop = (int (*)())dlsym(h, "roctxRangePop");
                if (r.push && r.pop) return r;
                r = Roctx{};
            }
        }
        std::fprintf(stderr, "pedoni_hip: PEDONI_ROCTX=1 but no roctx library could be loaded; ranges are off\n");
        return r;
    }();
    return api;
}
struct Range {
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

} // namespace

struct PedoniShard;

// Which instantiation of the queue force kernel a launch takes (kernels.hpp): the 94-SGPR build
// that holds 7 waves per SIMD, or the compiler's default register budget; SLOTS = candidates per
// lane and batch (the pair queue's depth).  PEDONI_FORCE_KERNEL = "s94:6", "default:8", ...
// overrides the by-size default (tools/slots_sweep.sh, tools/ab_repeat.sh).
enum class ForceBuild { BySize, S94, Default };
struct ForceChoice {
    ForceBuild build = ForceBuild::BySize;
    int slots = 0;
};

struct PedoniModel {
    int device = 0;
    PedoniShard* shard = nullptr;           // the shard driving this model, if any (shard.hpp)
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t side_stream = nullptr;      // interior rows of a split sharded tick
    hipEvent_t ev_sorted = nullptr, ev_interior = nullptr;
    hipEvent_t ev_tick[2] = {nullptr, nullptr}; // pedoni_hip_tick: device time of update_states
    PedoniOptions opt{};
    float size_x = 0, size_y = 0;
    Rng rng{12345};

    // field (field.rs:194-205)
    FieldView field{};
    float* d_distance_map = nullptr;
    float* d_pot = nullptr;              // n_maps potential maps, one after the other
    PedoniObstacle* d_obstacles = nullptr;
    uint32_t n_obstacles = 0;

    // neighbor grid (neighbor_grid.rs)
    GridView grid{};
    uint32_t n_cells = 0;
    int32_t band_lo = 0, band_hi = 0;

    // agents
    uint32_t cap = 0;
    float2* d_pos[2] = {nullptr, nullptr};
    float4* d_velx[2] = {nullptr, nullptr}; // {vx, vy, |v| * 0.1 (filled by the sort pass), desired speed}
    uint32_t* d_dest[2] = {nullptr, nullptr};
    int pv = 0; // buffer holding current pos / velx
    int vd = 0; // buffer holding current destination
    uint32_t* d_key = nullptr;
    uint32_t* d_slots = nullptr;
    uint32_t* d_scan_in = nullptr;  // cell_count (grid) or flags (no grid)
    uint32_t* d_cs[2] = {nullptr, nullptr}; // cell_start ping-pong (grid) / prefix (no grid, [0])
    int cs = 0;                     // d_cs[cs] = neighbor_grid_indices of the current order
    uint32_t* d_skey[2] = {nullptr, nullptr}; // packed (cy << 16 | cx) cell of each sorted agent
    int sk = 0;
    SortFlags* d_flags = nullptr;
    uint32_t tick_parity = 0;
    bool have_old = false;          // d_cs[cs] / d_skey[sk] describe the stored order
    bool keys_valid = false;        // d_key[base, live) already holds the next pass's keys (fused in K_FORCE)
    bool counts_dirty = false;      // d_scan_in holds per-cell counts of keys written since the last scan
    uint32_t scan_cap = 0;
    uint32_t* d_block_sums = nullptr;
    uint32_t block_sums_cap = 0;
    uint32_t* d_row_count = nullptr; // members per grid row (top level of the row scan)
    uint32_t* d_cell_flags = nullptr; // per-cell early-out flags (kernels.hpp CELL_FLAG_*); null with PEDONI_NO_CELL_FLAGS=1
    // heaviest tiles first (kernels.hpp build_tile_order): per-wave weights left by the last force launch, and the
    // workgroup order the sort pass builds from them for the next one; PEDONI_NO_TILE_ORDER=1 turns it off
    uint32_t* d_tile_weight = nullptr;
    uint32_t* d_tile_order = nullptr;
    uint32_t tile_order_blocks = 0;   // the force grid the current order was built for (0: none)
    bool tile_order_on = true;
    uint32_t* d_tickets = nullptr;   // [8 * TICKET_STRIDE] = place_kernel's workgroups-done counter; in front of it, the
                                     // diagnostics build's 8 tile-ticket words, TICKET_STRIDE apart
    bool tickets_fresh = false;      // zeroed by the place kernel and not drawn from since
    int force_persist = -1;          // PEDONI_FORCE_PERSIST: 0 never, 1 / 7, 6, 5 = the persistent form at that residency; -1 = by size
    uint32_t* d_live = nullptr; // device: [0] live agent count (absolute end index), [1] sticky status word
    uint32_t* h_pinned = nullptr;
    float2* d_acc = nullptr;
    uint32_t acc_cap = 0;

    // absolute indices into the agent arrays: [base, live) sorted agents, [live, gap_end) stale
    // slots, [gap_end, n_upper) appended since the last pass.  base = 0 unless the model is
    // one band of a sharded run: then [base - halo_cap, base) is the landing zone of the list
    // received from the band below (halo_unpack_kernel).
    uint32_t base = 0;
    uint32_t halo_cap = 0;
    HaloIn* d_halo = nullptr;
    // on-device periodic spawning
    SpawnerDev* d_spawners = nullptr;
    SpawnState* d_spawn_state = nullptr;
    uint32_t n_spawners = 0, spawn_cap = 0;
    uint32_t n_upper = 0; // host upper bound of the end of stored agents
    uint32_t gap_end = 0;
    bool sorted = false;  // cell_start matches the current pos buffer
    bool split_pending = false; // halo_tick_begin ran, halo_tick_end has not
    uint32_t ticks_since_tighten = 0; // device-stored appends since the host last read the count
    bool halo_keys_done = false; // halo_unpack_kernel keyed the exchanged agents of this pass
    bool force_simple = false; // PEDONI_FORCE_SIMPLE=1: one-lane-per-agent force kernel
    int ablate = 0;            // PEDONI_ABLATE bitmask: timing diagnostics only, results wrong
    uint32_t place_ablate = 0; // diagnostics build: place_kernel's switches (pedoni_hip_debug_set_ablate bits 8 and up)
    bool sort_general = false; // PEDONI_SORT_GENERAL=1: always take the atomic (general) sort form
    bool no_fuse_key = false;  // PEDONI_NO_FUSE_KEY=1: standalone K_KEY every tick
    bool xcd_remap = true;     // PEDONI_NO_XCD_REMAP=1: hardware block order
    unsigned long long* d_trace = nullptr; // PEDONI_FORCE_TRACE=1: per-phase cycle sums of the force kernel
    ForceChoice force_choice{}; // PEDONI_FORCE_KERNEL: build and queue depth of the force kernel (unset = by size)
    int force_group = -1;       // PEDONI_FORCE_GROUP: lanes per agent of the force kernel (1, 2, 4; unset = by size)
    int force_group_slots = 0;  // PEDONI_FORCE_GROUP_SLOTS: queue depth of the group kernel (4, 6, 8; unset = 6)

    // steady-state tick pair captured as a hipGraph (pedoni_hip_tick_n); see tick_graph()
    // one captured tick pair per tick parity (the ping-pong buffers a pair starts from): a run that
    // mixes eager ticks in (every n-th tick event-timed, n odd) starts its pairs on both
    struct TickGraph {
        hipGraphExec_t exec = nullptr;
        bool valid = false;
        uint32_t n_upper = 0, base = 0;
        int pv = 0, vd = 0, cs = 0, sk = 0;
        hipStream_t stream = nullptr;
    } graphs[2], long_graphs[3][2];
    // ... and, per parity, runs of 16 / 8 / 4 ticks in ONE graph launch each: between two graph launches the stream
    // idles ~8 us (tools/tick_timeline.sh) -- 4 us per tick with pairs, 0.5 with runs of 16.  A run is taken when that
    // many ticks ahead are neither event-timed nor beyond the call (PEDONI_GRAPH_TICKS: longest run allowed, 0 = pairs only)
    uint32_t graph_long = 16;
    void drop_graphs()
    {
        graphs[0].valid = graphs[1].valid = false;
        for (auto& lg : long_graphs) lg[0].valid = lg[1].valid = false;
    }
    bool use_graph = true;     // PEDONI_NO_GRAPH=1: always launch eagerly
    // edge-first force launch of a band (launch_force part 3; set by the shard driver, shard.hpp)
    uint32_t* edge_counter = nullptr;   // device word
    uint32_t* edge_flag = nullptr;      // device word
    uint32_t edge_seq = 0;
    // the next sort pass's first launch waits (in the kernel) for this word to reach this number: the
    // lists unpacked on the shard's communication stream (shard.hpp); consumed by that pass
    const uint32_t* scan_wait_flag = nullptr;
    uint32_t scan_wait_seq = 0;
    // ... riding on the scan's own workgroups only while their grid leaves the chip room for the other stream's
    // launches: half of what the device holds of them at once (PEDONI_SCAN_WAIT_ROWS_MAX overrides; tests)
    uint32_t scan_wait_rows_max = 0;

    // profiling
    uint32_t profile_mask = 0;
    uint32_t profile_every = 1;   // tick_n: time the kernels of every n-th tick only ...
    uint32_t profile_burst = 1;   // ... or of `burst` ticks in a row out of every n (pedoni_hip_profile_burst), counted from
    uint64_t profile_phase = 0;   // this tick number
    bool sampled(uint64_t t) const { return profile_mask != 0 && (t - profile_phase) % profile_every < profile_burst; }
    uint64_t tick_counter = 0;
    bool profile_now = true;      // false while tick_n runs a tick that is not sampled
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    PedoniKernelTimes times{};
};

void shard_detach_model(PedoniModel* m);   // shard.hpp

namespace {

int bind(PedoniModel* m)
{
    if (!m) return fail(PEDONI_E_INVALID, "null model");
    HIP_TRY(hipSetDevice(m->device));
    return PEDONI_OK;
}

int drain_events(PedoniModel* m)
{
    if (m->ev_used == 0) return PEDONI_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (size_t i = 0; i < m->ev_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, m->ev_pool[i].a, m->ev_pool[i].b));
        m->times.total_ms[m->ev_pool[i].kernel] += ms;
        m->times.launches[m->ev_pool[i].kernel] += 1;
    }
    m->ev_used = 0;
    return PEDONI_OK;
}

// RAII-less helper: records an event pair around a launch when profiling is on
struct Timed {
    PedoniModel* m;
    EventPair* ep = nullptr;
    int rc = PEDONI_OK;
    Range range;
    Timed(PedoniModel* m_, int kernel) : m(m_), range(kernel >= 0 ? KERNEL_NAMES[kernel] : "force_integrate (side stream)")
    {
        if (kernel < 0 || !m->profile_now || !((m->profile_mask >> kernel) & 1u)) return;
        if (m->ev_used == m->ev_pool.size()) {
            if (m->ev_pool.size() >= 8192) {
                rc = drain_events(m);
                if (rc) return;
            } else {
                EventPair p{};
                if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) {
                    rc = fail(PEDONI_E_HIP, "hipEventCreate failed");
                    return;
                }
                m->ev_pool.push_back(p);
            }
        }
        ep = &m->ev_pool[m->ev_used++];
        ep->kernel = kernel;
        hipEventRecord(ep->a, m->stream);
    }
    ~Timed()
    {
        if (ep) hipEventRecord(ep->b, m->stream);
    }
};

template <typename T> int dev_alloc(T** p, size_t n)
{
    HIP_TRY(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    return PEDONI_OK;
}

// Grows the agent arrays.  Every new buffer is allocated before anything is released, so a
// failed allocation leaves the model exactly as it was (and leaks nothing).
int ensure_capacity(PedoniModel* m, uint32_t need)
{
    if (need <= m->cap) return PEDONI_OK;
    uint64_t nc = std::max<uint64_t>(m->cap ? m->cap : 1024, 1024);
    while (nc < need) nc *= 2;
    // (the force kernel's pair queue packs a neighbour index into 26 bits beside the owner lane)
    if (nc > (1ull << 26)) return fail(PEDONI_E_INVALID, "agent capacity exceeds 2^26 (67 M) agents per GPU");
    const uint32_t ncap = (uint32_t)nc;
    const bool per_agent_scan = !m->opt.use_neighbor_grid; // the scan runs over per-agent flags
    const uint32_t nscan = ncap + 1, nsums = (nscan + SCAN_TILE - 1) / SCAN_TILE + 1;

    std::vector<void*> fresh;
    auto grab = [&](auto** p, size_t n) -> int {
        int rc = dev_alloc(p, n);
        if (rc == PEDONI_OK) fresh.push_back((void*)*p);
        return rc;
    };
    float2* npos[2] = {nullptr, nullptr};
    float4* nvel[2] = {nullptr, nullptr};
    uint32_t *ndest[2] = {nullptr, nullptr}, *nskey[2] = {nullptr, nullptr};
    uint32_t *nkey = nullptr, *nslots = nullptr, *nscan_in = nullptr, *ncs0 = nullptr, *nblock_sums = nullptr;
    uint32_t *ntile_weight = nullptr, *ntile_order = nullptr;
    const size_t n_tile_waves = (size_t)ncap / 64 + 8, n_tiles = (size_t)ncap / FORCE_THREADS + 8;
    int rc = PEDONI_OK;
    for (int k = 0; k < 2 && rc == PEDONI_OK; ++k) {
        rc = grab(&npos[k], ncap);
        if (rc == PEDONI_OK) rc = grab(&nvel[k], ncap);
        if (rc == PEDONI_OK) rc = grab(&ndest[k], ncap);
        if (rc == PEDONI_OK) rc = grab(&nskey[k], ncap);
    }
    if (rc == PEDONI_OK) rc = grab(&nkey, ncap);
    if (rc == PEDONI_OK) rc = grab(&nslots, ncap);
    if (rc == PEDONI_OK) rc = grab(&ntile_weight, n_tile_waves);
    if (rc == PEDONI_OK) rc = grab(&ntile_order, n_tiles);
    if (rc == PEDONI_OK && hipMemsetAsync(ntile_weight, 0, n_tile_waves * sizeof(uint32_t), m->stream) != hipSuccess)
        rc = fail(PEDONI_E_HIP, "ensure_capacity: hipMemsetAsync");
    if (per_agent_scan) {
        if (rc == PEDONI_OK) rc = grab(&nscan_in, nscan);
        if (rc == PEDONI_OK) rc = grab(&ncs0, nscan);
        if (rc == PEDONI_OK) rc = grab(&nblock_sums, nsums);
    }
    auto copy_over = [&]() -> int {
        if (!m->n_upper) return PEDONI_OK;
        const size_t n = m->n_upper;
        HIP_TRY(hipMemcpyAsync(npos[m->pv], m->d_pos[m->pv], n * sizeof(float2), hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(nvel[m->pv], m->d_velx[m->pv], n * sizeof(float4), hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(ndest[m->vd], m->d_dest[m->vd], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        return PEDONI_OK;
    };
    if (rc == PEDONI_OK) rc = copy_over();
    if (rc != PEDONI_OK) {
        for (void* p : fresh) hipFree(p);
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(m->stream)); // nothing in flight still reads the old arrays
    for (int k = 0; k < 2; ++k) {
        hipFree(m->d_pos[k]); hipFree(m->d_velx[k]); hipFree(m->d_dest[k]);
        hipFree(m->d_skey[k]);
        m->d_pos[k] = npos[k]; m->d_velx[k] = nvel[k]; m->d_dest[k] = ndest[k];
        m->d_skey[k] = nskey[k];
    }
    hipFree(m->d_key); hipFree(m->d_slots);
    m->d_key = nkey;
    m->d_slots = nslots;
    hipFree(m->d_tile_weight); hipFree(m->d_tile_order);
    m->d_tile_weight = ntile_weight;
    m->d_tile_order = ntile_order;
    m->tile_order_blocks = 0;
    if (per_agent_scan) {
        hipFree(m->d_scan_in); hipFree(m->d_cs[0]); hipFree(m->d_block_sums);
        m->d_scan_in = nscan_in;
        m->d_cs[0] = ncs0;
        m->d_block_sums = nblock_sums;
        m->scan_cap = nscan;
        m->block_sums_cap = nsums;
    }
    m->have_old = false; // the per-agent old-cell array did not survive the reallocation
    m->keys_valid = false;
    m->drop_graphs();
    m->cap = ncap;
    return PEDONI_OK;
}

// exclusive scan of in[0..n) -> out[0..n) (+ base), total -> out[n] and d_live: the
// three-launch form, used by the no-grid option path (per-agent survivor flags)
int run_scan(PedoniModel* m, uint32_t* in, uint32_t n, int zero_input, uint32_t* out)
{
    Timed t(m, PEDONI_K_SCAN);
    if (t.rc) return t.rc;
    uint32_t n_blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (n_blocks == 0) n_blocks = 1;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(n_blocks), dim3(SCAN_THREADS), 0, m->stream,
                       in, n, m->d_block_sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, m->stream, m->d_block_sums,
                       n_blocks, m->base, out + n, m->d_live);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(n_blocks), dim3(SCAN_THREADS), 0, m->stream,
                       in, n, m->d_block_sums, out, zero_input);
    HIP_TRY(hipGetLastError());
    return PEDONI_OK;
}

// cell counts of grid rows [row0, row1) -> cell_start (neighbor_grid_indices), one launch
int run_row_scan(PedoniModel* m, int32_t row0, int32_t row1, uint32_t* out, uint32_t limit)
{
    const bool sharded_wait = m->scan_wait_flag != nullptr;
    if (sharded_wait && (uint32_t)(row1 - row0) > m->scan_wait_rows_max) {
        // One spinning workgroup per row would fill the chip's wave slots (a band of ~2000 rows does) and
        // keep out the communication stream's unpack and post launches -- the very ones that store the word:
        // stream priority does not preempt resident waves.  Such a band waits in ONE wave, ahead of the scan.
        hipLaunchKernelGGL(edge_wait_kernel, dim3(1), dim3(64), 0, m->stream, m->scan_wait_flag, m->scan_wait_seq,
                           m->d_live + 1, 500000000ull);
        HIP_TRY(hipGetLastError());
        m->scan_wait_flag = nullptr;
    }
    Timed t(m, PEDONI_K_SCAN);
    if (t.rc) return t.rc;
    if (sharded_wait)
        hipLaunchKernelGGL(scan_rows_kernel<true>, dim3((uint32_t)(row1 - row0)), dim3(SCAN_THREADS), 0, m->stream,
                           m->d_scan_in, m->d_row_count, row0, m->grid.cols, m->base, out, m->d_live, limit,
                           m->d_live + 1, m->scan_wait_flag, m->scan_wait_seq);
    else
        hipLaunchKernelGGL(scan_rows_kernel<false>, dim3((uint32_t)(row1 - row0)), dim3(SCAN_THREADS), 0, m->stream,
                           m->d_scan_in, m->d_row_count, row0, m->grid.cols, m->base, out, m->d_live, limit,
                           m->d_live + 1, (const uint32_t*)nullptr, 0u);
    m->scan_wait_flag = nullptr;
    HIP_TRY(hipGetLastError());
    return PEDONI_OK;
}

inline uint32_t blocks_for(uint32_t n, uint32_t bs) { return std::max(1u, (n + bs - 1) / bs); }

// Which instantiation of the grid-path force kernel a launch over n agents takes: lanes per agent
// (group), build, queue depth.  One place for the rule; launch_force follows it and
// pedoni_hip_force_kernel_info reports it (bench.py prices a run with the profile of THAT kernel).
struct ForcePlan { int group; ForceBuild build; int slots; };
int group_by_size(uint32_t n);
// whole_array false = the interior rows of a split sharded tick (one-lane kernel); the EDGE rows' launch
// (a few thousand agents: bound by one wave's critical path) is planned like a small crowd of its size
inline ForcePlan plan_force(const PedoniModel* m, uint32_t n, bool whole_array)
{
    ForcePlan p{1, m->force_choice.build, m->force_choice.slots};
    if (p.build == ForceBuild::BySize) {
        // 6 candidate slots per lane and batch in the 94-SGPR, 7-waves-per-SIMD build from 4e5 agents up
        // (96.2 us against 100.5 us for the default build at N = 1e6, round 2); smaller crowds, whose
        // waves are few anyway, run the default build with 8-slot batches -- or the group kernel
        const ForcePlan by_n = n >= 400000u ? ForcePlan{1, ForceBuild::S94, 6} : ForcePlan{1, ForceBuild::Default, 8};
        p.build = by_n.build; p.slots = by_n.slots;
        p.group = m->force_group > 0 ? m->force_group : group_by_size(n);
        if (!whole_array) p.group = 1;
        if (p.group > 1) {
            p.build = ForceBuild::Default;
            p.slots = m->force_group_slots == 4 || m->force_group_slots == 6 || m->force_group_slots == 8
                          ? m->force_group_slots : (p.group == 2 ? 8 : 6);   // (C2: 8 slots 49.9 us, 6 slots 50.5)
        }
    }
    return p;
}

// Lanes per agent of the force kernel by crowd size (kernels.hpp force_kernel_queue_group), from
// tools/group_n_sweep.sh on MI355X (profiles/r03_group_n_sweep.txt; tick, us, G = 1 / 2 / 4):
// N = 25 000: 34.2 / 27.1 / 24.7; 50 000: 34.7 / 28.6 / 29.1; 1e5: 34.1 / 32.0 / 34.9; 2e5: 43.6 / 42.8 / 47.9;
// 3e5: 50.7 / 52.0 / 61.3 (rho = 1); rho = 2.5: 1e5: 51.1 / 46.7 / 46.7, 3e5: 81.3 / 76.5 / 82.1.
// Small launches are bound by their heaviest wave's critical path, which the group form shortens;
// from ~3e5 agents on the SIMDs are busy and the group's redundant per-agent work costs more than
// the shorter path brings.
int group_by_size(uint32_t n) { return n < 40000u ? 4 : (n < 250000u ? 2 : 1); }

// heaviest tiles first: only for the launch the order is built for -- every sorted agent of an unsharded model
// through a one-lane-per-agent kernel in the XCD-contiguous order (a band's edge-first launch has an order of
// its own; small crowds run the group kernel, whose launch is a single generation)
bool tile_order_wanted(const PedoniModel* m, uint32_t n)
{
    if (!m->tile_order_on || !m->d_tile_order || !m->xcd_remap || m->halo_cap || m->force_simple || !m->opt.use_neighbor_grid || n == 0)
        return false;
    if (blocks_for(n, FORCE_THREADS) / 8u + 1u > TILE_CHUNK_MAX) return false;      // (more tiles per XCD than the builder holds)
    return plan_force(m, n, true).group == 1;
}

} // namespace
#ifdef PEDONI_DIAGNOSTICS
#include "host_diag.hpp"
#endif
namespace {

// sfm.rs:58-88 on the device
int sort_despawn(PedoniModel* m)
{
    Range pass("spawn_pedestrians: sort/despawn pass (sfm.rs:58-88)");
    uint32_t n_total = m->n_upper;
    const uint32_t i0 = m->base - m->halo_cap;     // first index a pass may have to look at
    const uint32_t n_threads = n_total - i0;
    const int src = m->pv, dst = 1 - m->pv, vsrc = m->vd, vdst = 1 - m->vd;
    const uint32_t bs = 256;
    if (n_threads == 0) {
        // nothing stored: live count 0, cell_start all zero (only reachable with base == 0)
        HIP_TRY(hipMemsetAsync(m->d_live, 0, sizeof(uint32_t), m->stream));
        if (m->opt.use_neighbor_grid)
            HIP_TRY(hipMemsetAsync(m->d_cs[m->cs], 0, (size_t)(m->n_cells + 1) * sizeof(uint32_t),
                                   m->stream));
        m->have_old = false;
        m->sorted = true;
        return PEDONI_OK;
    }
    if (m->opt.use_neighbor_grid) {
        const int cs_old = m->cs, cs_new = 1 - m->cs, sk_old = m->sk, sk_new = 1 - m->sk;
        const uint32_t parity = m->tick_parity & 1u;
        // the gather form needs last tick's order and 16-bit cell coordinates
        const bool packable = m->grid.rows <= 0xffff && m->grid.cols <= 0xffff;
        const int force_general = (!m->have_old || !packable || m->sort_general) ? 1 : 0;
        const BandView band{m->band_lo, m->band_hi, m->halo_cap ? 1 : 0};
        // a band only ever touches the cells of rows lo-1 .. hi: scan just those
        const int32_t row0 = std::max(m->band_lo - 1, 0), row1 = std::min(m->band_hi + 1, m->grid.rows);
        SoA soa{m->d_pos[src], m->d_velx[src], m->d_dest[vsrc],
                m->d_pos[dst], m->d_velx[dst], m->d_dest[vdst], m->d_skey[sk_new],
                m->opt.math_mode == PEDONI_MATH_FAST ? 1 : 0};
        // K_KEY only when something needs a key: every stored agent (first pass, general form), or the
        // agents stored since the last update (appended, exchanged).  A steady-state tick launches
        // nothing here -- and records no event pair for it either.
        const bool key_all = !m->keys_valid || force_general;
        const bool key_halo = !key_all && m->halo_cap && !m->halo_keys_done;
        const bool key_appended = !key_all && n_total > m->gap_end && !m->halo_keys_done;
        if (key_all || key_halo || key_appended) {
            if (m->scan_wait_flag) {
                // (a pass that keys agents first: the wait cannot ride on the scan, it gets a launch of its own)
                hipLaunchKernelGGL(edge_wait_kernel, dim3(1), dim3(64), 0, m->stream, m->scan_wait_flag, m->scan_wait_seq,
                                   m->d_live + 1, 500000000ull);
                m->scan_wait_flag = nullptr;
                HIP_TRY(hipGetLastError());
            }
            Timed t(m, PEDONI_K_BIN);
            if (t.rc) return t.rc;
            if (key_all) {
                // every stored agent needs its key (and its cell's count: drop what a fused
                // update may already have accumulated for keys that are now recomputed)
                if (m->counts_dirty) {
                    HIP_TRY(hipMemsetAsync(m->d_scan_in, 0, (size_t)(m->n_cells + 1) * sizeof(uint32_t),
                                           m->stream));
                    HIP_TRY(hipMemsetAsync(m->d_row_count, 0, (size_t)(m->grid.rows + 1) * sizeof(uint32_t),
                                           m->stream));
                }
                hipLaunchKernelGGL(key_kernel, dim3(blocks_for(n_threads, bs)), dim3(bs), 0, m->stream,
                                   m->d_pos[src], m->d_dest[vsrc], i0, n_total, m->base, m->d_live,
                                   m->gap_end, m->d_halo, m->field, m->grid, m->band_lo, m->band_hi,
                                   m->d_skey[sk_old], force_general, parity, m->d_flags, m->d_key,
                                   m->d_scan_in, m->d_row_count);
            } else {
                // own agents got their keys (and counts) from the last update_states; only agents
                // stored since then are keyed here (exchanged lists: by halo_unpack_kernel)
                if (key_halo)
                    hipLaunchKernelGGL(key_kernel, dim3(blocks_for(m->halo_cap, bs)), dim3(bs), 0,
                                       m->stream, m->d_pos[src], m->d_dest[vsrc], i0, m->base, m->base,
                                       m->d_live, m->gap_end, m->d_halo, m->field, m->grid, m->band_lo,
                                       m->band_hi, m->d_skey[sk_old], 0, parity, m->d_flags, m->d_key,
                                       m->d_scan_in, m->d_row_count);
                if (key_appended)
                    hipLaunchKernelGGL(key_kernel, dim3(blocks_for(n_total - m->gap_end, bs)), dim3(bs),
                                       0, m->stream, m->d_pos[src], m->d_dest[vsrc], m->gap_end, n_total,
                                       m->base, m->d_live, m->gap_end, m->d_halo, m->field, m->grid,
                                       m->band_lo, m->band_hi, m->d_skey[sk_old], 0, parity, m->d_flags,
                                       m->d_key, m->d_scan_in, m->d_row_count);
            }
        }
        TRY(run_row_scan(m, row0, row1, m->d_cs[cs_new], n_total));
        m->counts_dirty = false;
        // The reorder launch (general form: in-cell order by previous index) is needed when the host
        // KNOWS agents take the general form: first pass / forced, agents stored since the last update
        // (their keys raise the device flag), a band's boundary rows, device spawning.  Otherwise --
        // the steady-state tick -- it is not launched at all (it used to be: a 128-block kernel that read
        // one flag and left, 5-6 us per tick), and place_kernel's last workgroup covers the one case
        // only the device can know of (an agent that moved more than one cell).
        const bool host_knows_general = force_general || n_total > m->gap_end || m->halo_cap != 0 || m->n_spawners != 0 ||
                                        m->halo_keys_done;
        {
            Timed t(m, PEDONI_K_SLOT);
            if (t.rc) return t.rc;
            // (workgroups of 64 ... 1024 threads: the launch takes the same 17.6-18.9 us, tools/place_probe.sh)
            uint32_t* const tickets = m->force_persist > 0 ? m->d_tickets : nullptr;
            uint32_t* const done_count = host_knows_general ? nullptr : m->d_tickets + 8 * TICKET_STRIDE;
            HaloIn* const consumed = (m->halo_cap || m->n_spawners) ? m->d_halo : nullptr;
            // the workgroup order of the force launch this pass is followed by (whole-array launches of the
            // one-lane kernels of an unsharded model: plan_force), built from the last launch's weights
            TileOrder tiles{nullptr, nullptr, 0u};
            m->tile_order_blocks = 0;
            if (tile_order_wanted(m, n_total - m->base)) {
                tiles = TileOrder{m->d_tile_weight, m->d_tile_order, blocks_for(n_total - m->base, FORCE_THREADS)};
                m->tile_order_blocks = tiles.n_blocks;
            }
#ifdef PEDONI_DIAGNOSTICS
            // (the place kernel with parts switched off, and the dispatch-cost probes behind it: host_diag.hpp)
            if (m->place_ablate)
                diag_launch_place(m, blocks_for(n_threads, bs), bs, i0, n_total, band, cs_old, cs_new, parity, soa, consumed, row0, row1,
                                  tickets, done_count);
            else
#endif
            hipLaunchKernelGGL(place_kernel, dim3(blocks_for(n_threads, bs)), dim3(bs), 0,
                               m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old],
                               m->d_cs[cs_new], m->d_flags, parity, m->d_scan_in, soa, m->d_slots,
                               consumed, m->d_row_count, row0, row1, m->d_live + 1, tickets, done_count, tiles);
            m->tickets_fresh = true;
        }
        if (host_knows_general) {
            Timed t(m, PEDONI_K_REORDER);
            if (t.rc) return t.rc;
            // grid-stride kernel: a band in steady state has only the agents of its four boundary rows in
            // general form (small grid); a whole pass in general form gets the full grid
            const uint32_t reorder_blocks = std::min(blocks_for(n_threads, bs), force_general ? 1024u : 128u);
            hipLaunchKernelGGL(reorder_kernel, dim3(reorder_blocks), dim3(bs), 0,
                               m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old],
                               m->d_cs[cs_new], m->d_slots, m->d_flags, parity, m->d_scan_in, soa);
        }
        m->cs = cs_new;
        m->sk = sk_new;
        m->tick_parity += 1;
        m->have_old = true;
        m->keys_valid = false; // consumed
        m->halo_keys_done = false;
    } else {
        {
            Timed t(m, PEDONI_K_BIN);
            if (t.rc) return t.rc;
            hipLaunchKernelGGL(flag_kernel, dim3(blocks_for(n_total, bs)), dim3(bs), 0, m->stream,
                               m->d_pos[src], m->d_dest[vsrc], n_total, m->d_live, m->gap_end,
                               m->field, m->d_scan_in);
        }
        // d_key keeps the flags: the scan may not zero what compact still reads
        HIP_TRY(hipMemcpyAsync(m->d_key, m->d_scan_in, (size_t)n_total * sizeof(uint32_t),
                               hipMemcpyDeviceToDevice, m->stream));
        TRY(run_scan(m, m->d_scan_in, n_total, /*zero_input=*/0, m->d_cs[0]));
        {
            Timed t(m, PEDONI_K_REORDER);
            if (t.rc) return t.rc;
            hipLaunchKernelGGL(compact_kernel, dim3(blocks_for(n_total, bs)), dim3(bs), 0,
                               m->stream, m->d_key, m->d_cs[0], n_total, m->d_pos[src],
                               m->d_velx[src], m->d_dest[vsrc], m->d_pos[dst], m->d_velx[dst],
                               m->d_dest[vdst]);
        }
    }
    HIP_TRY(hipGetLastError());
    m->pv = dst;
    m->vd = vdst;
    m->gap_end = m->n_upper; // every stored agent is now either live (< *d_live) or stale
    // (the device-stored agents are marked consumed by place_kernel; the no-grid path has none)
    m->sorted = true;
    return PEDONI_OK;
}

ForceArgs force_args(PedoniModel* m, float2* acc_out)
{
    ForceArgs a{};
    a.pos = m->d_pos[m->pv];
    a.velx = m->d_velx[m->pv];
    a.dest = m->d_dest[m->vd];
    a.pos_out = acc_out ? nullptr : m->d_pos[1 - m->pv];
    a.velx_out = acc_out ? nullptr : m->d_velx[1 - m->pv];
    a.acc_out = acc_out;
    a.live_count = m->d_live;
    a.base = m->base;
    a.cell_start = m->d_cs[m->cs];
    a.obstacles = m->d_obstacles;
    a.n_obstacles = m->n_obstacles;
    a.field = m->field;
    a.grid = m->grid;
    a.band_lo = m->band_lo;
    a.band_hi = m->band_hi;
    a.use_grid = m->opt.use_neighbor_grid;
    a.use_distance_map = m->opt.use_distance_map;
    a.ablate = m->ablate;
    a.trace = m->d_trace;
    a.xcd_remap = m->xcd_remap ? 1 : 0;
    a.seg_row[0][0] = -1; a.seg_row[0][1] = a.seg_row[1][0] = a.seg_row[1][1] = 0;
    a.clear_stale = 0;
    a.error_word = &m->d_halo->error;
    // fuse the next pass's K_KEY when this launch integrates (queue kernel, grid mode)
    const bool fuse = !acc_out && m->opt.use_neighbor_grid && !m->force_simple && !m->no_fuse_key &&
                      m->grid.rows <= 0xffff && m->grid.cols <= 0xffff;
    a.key_next = fuse ? m->d_key : nullptr;
    a.cell_count = m->d_scan_in;
    a.row_count = m->d_row_count;
    a.key_end = m->n_upper;
    a.flags = m->d_flags;
    a.parity_next = m->tick_parity & 1u;
    a.cell_flags = m->d_cell_flags;
    return a;
}

// part: 0 = every sorted agent; 3 = the same in ONE launch whose first workgroups take the edge rows and
// signal their completion (force_kernel_queue_edge_first); 1 = the rows next to the band's edges (ghost rows, which
// are only NaN-marked, and the two owned rows beside each); 2 = the interior rows.  Parts 1
// and 2 together equal part 0; they exist so that a sharded tick can pack and send its
// boundary agents while the interior is still being computed.
int launch_force(PedoniModel* m, float2* acc_out, int part = 0, hipStream_t on = nullptr)
{
    hipStream_t stream = on ? on : m->stream;
    uint32_t n = m->n_upper - m->base;
    if (n == 0) return PEDONI_OK;
    uint32_t bs = m->opt.gpu_work_size > 0 ? (uint32_t)m->opt.gpu_work_size : 256u;
    ForceArgs a = force_args(m, acc_out);
    if (part != 0) {
        const int32_t rows = m->grid.rows;
        const int32_t lo_a = std::max(m->band_lo - 1, 0), lo_b = std::min(m->band_lo + 2, m->band_hi);
        const int32_t hi_a = std::max(m->band_hi - 2, lo_b), hi_b = std::min(m->band_hi + 1, rows);
        if (part == 3) {
            if (!m->edge_flag || !m->edge_counter) return fail(PEDONI_E_INVALID, "edge-first force launch without its signal word");
            a.edge_row[0] = lo_b; a.edge_row[1] = hi_a;
            a.edge_counter = m->edge_counter; a.edge_flag = m->edge_flag; a.edge_seq = m->edge_seq;
            // where the edge rows should be (block_order.hpp; a hint: the kernel finds the real ones itself).  The
            // live agents end about where the host's bound stood before the unpacks since its last reading
            // of the count each added their capacity
            const uint32_t lists = std::max(1u, (m->band_lo > 0 ? 1u : 0u) + (m->band_hi < rows ? 1u : 0u));
            const EdgeHint h = edge_first_hint(n, FORCE_THREADS, m->halo_cap,
                                               (uint64_t)(m->ticks_since_tighten + 1u) * m->halo_cap * lists);
            a.edge_blocks[0] = h.e_lo; a.edge_blocks[1] = h.e_hi; a.edge_tile_hi = h.t_hi;
        } else if (part == 1) {
            a.seg_row[0][0] = lo_a; a.seg_row[0][1] = lo_b;
            a.seg_row[1][0] = hi_a; a.seg_row[1][1] = hi_b;
            n = std::min(n, 8u * std::max(m->halo_cap, 1u)); // 6 rows, each <= ~0.7 halo_cap
        } else {
            a.seg_row[0][0] = lo_b; a.seg_row[0][1] = hi_a;
            a.seg_row[1][0] = a.seg_row[1][1] = hi_a;
            a.clear_stale = 1;
        }
    }
    if (part == 0 && !acc_out && tile_order_wanted(m, n)) {
        // (the order the sort pass built is this launch's if it was built for this grid; the weights this launch
        // leaves are the next pass's input either way)
        if (m->tile_order_blocks == blocks_for(n, FORCE_THREADS)) a.tile_order = m->d_tile_order;
        a.tile_weight = m->d_tile_weight;
    }
    Timed t(m, on ? -1 : PEDONI_K_FORCE);   // the side-stream launch is not event-timed
    if (t.rc) return t.rc;
    const bool fast = m->opt.math_mode == PEDONI_MATH_FAST;
    if (m->opt.use_neighbor_grid && !m->force_simple) {
        dim3 grid(blocks_for(n, FORCE_THREADS)), block(FORCE_THREADS);
        // 6 candidate slots per lane and batch (12-byte queue entries: 18 KB LDS per block).  For
        // large crowds the 94-SGPR, 7-waves-per-SIMD build of the kernel (kernels.hpp
        // force_kernel_queue_s94): 96.2 us against 100.5 us at N = 1e6, exact mode; small crowds
        // (few waves per SIMD anyway) run the default build with 8-slot batches.  PEDONI_FORCE_SLOTS overrides:
        // 4 / 5 (s94), 6, 8, 15 (5 slots, default SGPRs), 16 / 18 (s94 with 6 / 8 slots).
        const ForcePlan c = plan_force(m, n, part != 2);
        if (part == 3) {
            // (the caller has checked that the plan is a one-lane-per-agent build: edge_first_ready)
            if (c.group > 1 || (c.build == ForceBuild::S94 && c.slots != 6) || (c.build != ForceBuild::S94 && c.slots != 8))
                return fail(PEDONI_E_INVALID, "edge-first force launch: no such build of the kernel");
            if (c.build == ForceBuild::S94) {
                if (fast) hipLaunchKernelGGL((force_kernel_queue_edge_first_s94<1, 6>), grid, block, 0, stream, a);
                else hipLaunchKernelGGL((force_kernel_queue_edge_first_s94<0, 6>), grid, block, 0, stream, a);
            } else {
                if (fast) hipLaunchKernelGGL((force_kernel_queue_edge_first<1, 8>), grid, block, 0, stream, a);
                else hipLaunchKernelGGL((force_kernel_queue_edge_first<0, 8>), grid, block, 0, stream, a);
            }
            HIP_TRY(hipGetLastError());
            return PEDONI_OK;
        }
#ifdef PEDONI_DIAGNOSTICS
        // (diagnostic instantiations -- per-phase trace, ablation switches, one wave per workgroup, the
        // persistent forms -- selected by PEDONI_FORCE_TRACE / PEDONI_ABLATE / PEDONI_FORCE_PERSIST: host_diag.hpp)
        {
            int rc = PEDONI_OK;
            if (diag_launch_force(m, a, c, n, grid, block, stream, fast, part, on != nullptr, &rc)) return rc;
        }
#endif
        auto launch = [&](auto exact_kernel, auto fast_kernel) {
            if (fast) hipLaunchKernelGGL(fast_kernel, grid, block, 0, stream, a);
            else hipLaunchKernelGGL(exact_kernel, grid, block, 0, stream, a);
        };
        // small crowds: G lanes per agent (kernels.hpp force_kernel_queue_group) -- whole-array launches only
        const int group = c.group;
        if (group > 1) {
            grid = dim3(blocks_for(n, FORCE_THREADS / (uint32_t)group));
            const int gs = c.slots;
            if (group == 2) {
                if (gs == 8) launch(force_kernel_queue_group<0, 8, 2>, force_kernel_queue_group<1, 8, 2>);
                else if (gs == 4) launch(force_kernel_queue_group<0, 4, 2>, force_kernel_queue_group<1, 4, 2>);
                else launch(force_kernel_queue_group<0, 6, 2>, force_kernel_queue_group<1, 6, 2>);
            } else {
                if (gs == 8) launch(force_kernel_queue_group<0, 8, 4>, force_kernel_queue_group<1, 8, 4>);
                else if (gs == 4) launch(force_kernel_queue_group<0, 4, 4>, force_kernel_queue_group<1, 4, 4>);
                else launch(force_kernel_queue_group<0, 6, 4>, force_kernel_queue_group<1, 6, 4>);
            }
            HIP_TRY(hipGetLastError());
            return PEDONI_OK;
        }
        if (c.build == ForceBuild::S94) {
            switch (c.slots) {
            case 4: launch(force_kernel_queue_s94<0, 4>, force_kernel_queue_s94<1, 4>); break;
            case 5: launch(force_kernel_queue_s94<0, 5>, force_kernel_queue_s94<1, 5>); break;
            case 8: launch(force_kernel_queue_s94<0, 8>, force_kernel_queue_s94<1, 8>); break;
            default: launch(force_kernel_queue_s94<0, 6>, force_kernel_queue_s94<1, 6>);
            }
        } else {
            switch (c.slots) {
            case 5: launch(force_kernel_queue<0, 5>, force_kernel_queue<1, 5>); break;
            case 6: launch(force_kernel_queue<0, 6>, force_kernel_queue<1, 6>); break;
            default: launch(force_kernel_queue<0, 8>, force_kernel_queue<1, 8>);
            }
        }
    } else {
        if (part != 0) return fail(PEDONI_E_INVALID, "row-segment force launch needs the queue kernel");
        dim3 grid(blocks_for(n, bs)), block(bs);
        if (fast) hipLaunchKernelGGL(force_kernel_simple<1>, grid, block, 0, stream, a);
        else hipLaunchKernelGGL(force_kernel_simple<0>, grid, block, 0, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return PEDONI_OK;
}

// can the band's next force launch take the edge-first form?  (a one-lane-per-agent build of the
// queue kernel in its default batch size, something to launch, the signal word in place)
bool edge_first_ready(PedoniModel* m)
{
    if (!m->edge_flag || !m->edge_counter || m->n_upper <= m->base || !m->opt.use_neighbor_grid || m->force_simple) return false;
#ifdef PEDONI_DIAGNOSTICS
    if (m->d_trace || m->ablate || m->force_persist) return false;
#endif
    const ForcePlan c = plan_force(m, m->n_upper - m->base, true);
    return c.group == 1 && ((c.build == ForceBuild::S94 && c.slots == 6) || (c.build != ForceBuild::S94 && c.slots == 8));
}

void after_update(PedoniModel* m)
{
    if (m->n_upper > m->base) m->pv = 1 - m->pv;
    m->keys_valid = m->n_upper > m->base && m->opt.use_neighbor_grid && !m->force_simple &&
                    !m->no_fuse_key && m->grid.rows <= 0xffff && m->grid.cols <= 0xffff;
    if (m->keys_valid) m->counts_dirty = true;
    m->sorted = false;
}

// sfm.rs:91-255 on the device
int update_states(PedoniModel* m)
{
    if (!m->sorted)
        return fail(PEDONI_E_INVALID,
                    "update_states needs the sort/despawn pass of spawn_pedestrians first "
                    "(Simulator::tick order, lib.rs:85,90)");
    Range pass("update_states (sfm.rs:91-255)");
    TRY(launch_force(m, nullptr));
    after_update(m);
    return PEDONI_OK;
}

// the sticky device status word (kernels.hpp STATUS_*): no host read of device state
// succeeds while a bit is set
int check_status(uint32_t status)
{
    if (status & STATUS_SCAN_MISMATCH)
        return fail(PEDONI_E_HIP, "device status: a grid row's cell counts do not add up to its row count "
                                  "(neighbor_grid_indices would be wrong)");
    if (status & STATUS_LIVE_OVERFLOW)
        return fail(PEDONI_E_CAPACITY, "device status: more live agents than the host's bound of the arrays "
                                       "(received lists larger than the reserved capacity?)");
    if (status & STATUS_EDGE_WAIT)
        return fail(PEDONI_E_HIP, "device status: the overlapped band tick waited in vain for its force launch's edge rows "
                                  "(the lists sent that tick may be incomplete)");
    if (status & STATUS_FIELD_SLICE)
        return fail(PEDONI_E_CAPACITY, "device status: an agent sampled a field-map row outside the rows uploaded "
                                       "for this band (pedoni_hip_create_rows: widen the row range)");
    if (status) return fail(PEDONI_E_HIP, "device status word is set: " + std::to_string(status));
    return PEDONI_OK;
}

int sync_live_count(PedoniModel* m, uint32_t* out)
{
    HIP_TRY(hipMemcpyAsync(m->h_pinned, m->d_live, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                           m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    uint32_t live = m->h_pinned[0]; // absolute end index
    TRY(check_status(m->h_pinned[1]));
    if (m->n_spawners) { // a tick that spawned more than max_per_tick must not pass silently
        HaloIn h{};
        HIP_TRY(hipMemcpy(&h, m->d_halo, sizeof h, hipMemcpyDeviceToHost));
        if (h.error & 8u) return fail(PEDONI_E_CAPACITY, "a tick spawned more agents than max_per_tick");
    }
    // tighten the host bound when nothing has been appended since the last pass
    if (m->gap_end == m->n_upper) m->n_upper = m->gap_end = live;
    *out = live - m->base;
    return PEDONI_OK;
}

int append(PedoniModel* m, const float* pos_xy, const uint32_t* destination,
           const float* desired_speed, const float* vel_xy, uint32_t n)
{
    if (n == 0) return PEDONI_OK;
    if (!pos_xy || !destination) return fail(PEDONI_E_INVALID, "append: null pos/destination");
    if ((uint64_t)m->n_upper + n > 0xfffffff0ull)
        return fail(PEDONI_E_INVALID, "append: too many agents");
    TRY(ensure_capacity(m, m->n_upper + n));
    std::vector<float> v0(n);
    if (desired_speed) {
        std::memcpy(v0.data(), desired_speed, n * sizeof(float));
    } else {
        SpawnState st{};
        if (m->n_spawners) { // the desired-speed stream lives on the device: continue it
            HIP_TRY(hipMemcpyAsync(&st, m->d_spawn_state, sizeof st, hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            m->rng.s = st.rng_v0;
        }
        for (uint32_t i = 0; i < n; ++i) v0[i] = m->rng.normal_approx(1.34f, 0.26f); // sfm.rs:54
        if (m->n_spawners) {
            st.rng_v0 = m->rng.s;
            HIP_TRY(hipMemcpyAsync(m->d_spawn_state, &st, sizeof st, hipMemcpyHostToDevice, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
        }
    }
    size_t at = m->n_upper;
    HIP_TRY(hipMemcpyAsync(m->d_pos[m->pv] + at, pos_xy, n * sizeof(float2), hipMemcpyHostToDevice,
                           m->stream));
    // {vx, vy, (|v| * 0.1: filled by the sort pass), desired speed}; velocity 0 when not given (:53)
    std::vector<float4> velx(n);
    for (uint32_t i = 0; i < n; ++i)
        velx[i] = make_float4(vel_xy ? vel_xy[2 * i] : 0.0f, vel_xy ? vel_xy[2 * i + 1] : 0.0f, 0.0f, v0[i]);
    HIP_TRY(hipMemcpyAsync(m->d_velx[m->pv] + at, velx.data(), n * sizeof(float4), hipMemcpyHostToDevice,
                           m->stream));
    HIP_TRY(hipMemcpyAsync(m->d_dest[m->vd] + at, destination, n * sizeof(uint32_t),
                           hipMemcpyHostToDevice, m->stream));
    // pageable host buffers: the copies above are staged synchronously by the runtime, but
    // velx is a local -- make sure it has been consumed before it goes out of scope
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->n_upper += n;
    m->sorted = false;
    return PEDONI_OK;
}

} // namespace

// =========================================================================================
extern "C" {

const char* pedoni_hip_last_error(void) { return g_last_error.c_str(); }

const char* pedoni_hip_kernel_name(int32_t k)
{
    return (k >= 0 && k < PEDONI_N_KERNELS) ? KERNEL_NAMES[k] : "?";
}

void pedoni_hip_default_options(PedoniOptions* opt)
{
    if (!opt) return;
    std::memset(opt, 0, sizeof(*opt));
    opt->neighbor_grid_unit = 1.4f; // lib.rs:128
    opt->field_grid_unit = 0.25f;   // lib.rs:129
    opt->use_neighbor_grid = 1;     // lib.rs:130
    opt->use_distance_map = 1;      // lib.rs:131
    opt->gpu_work_size = 0;         // lib.rs:132 has 64; 0 = library default
    opt->math_mode = PEDONI_MATH_EXACT;
    opt->seed = 12345;
    opt->initial_capacity = 0;
}

int pedoni_hip_device_count(int32_t* n)
{
    if (!n) return fail(PEDONI_E_INVALID, "null out");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(PEDONI_E_NO_DEVICE, hipGetErrorString(e)); }
    *n = c;
    return PEDONI_OK;
}

int pedoni_hip_create(const PedoniOptions* opt, float size_x, float size_y,
                      const float* distance_map, const float* const* potential_maps,
                      uint32_t n_maps, uint32_t field_rows, uint32_t field_cols, float field_unit,
                      const PedoniObstacle* obstacles, uint32_t n_obstacles, int device,
                      PedoniModel** out)
{
    return pedoni_hip_create_rows(opt, size_x, size_y, distance_map, potential_maps, n_maps, field_rows, field_cols,
                                  field_unit, obstacles, n_obstacles, device, 0, field_rows, out);
}

int pedoni_hip_create_rows(const PedoniOptions* opt, float size_x, float size_y,
                           const float* distance_map, const float* const* potential_maps,
                           uint32_t n_maps, uint32_t field_rows, uint32_t field_cols, float field_unit,
                           const PedoniObstacle* obstacles, uint32_t n_obstacles, int device,
                           uint32_t map_row_begin, uint32_t map_row_end, PedoniModel** out)
{
    if (!opt || !out) return fail(PEDONI_E_INVALID, "create: null options/out");
    if (!distance_map || (n_maps && !potential_maps))
        return fail(PEDONI_E_INVALID, "create: null field maps");
    if (field_rows == 0 || field_cols == 0 || field_rows > 0x7fffffffu || field_cols > 0x7fffffffu)
        return fail(PEDONI_E_INVALID, "create: bad field shape");
    if (map_row_begin >= map_row_end || map_row_end > field_rows)
        return fail(PEDONI_E_INVALID, "create: bad map row range");
    if (n_obstacles && !obstacles) return fail(PEDONI_E_INVALID, "create: null obstacles");
    if (opt->gpu_work_size < 0 || opt->gpu_work_size > 1024 || (opt->gpu_work_size % 64) != 0)
        return fail(PEDONI_E_INVALID, "create: gpu_work_size must be a multiple of 64 <= 1024");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(PEDONI_E_NO_DEVICE, "create: no HIP device (this backend has no CPU fallback)");
    if (device < 0 || device >= n_dev) return fail(PEDONI_E_INVALID, "create: bad device index");
    HIP_TRY(hipSetDevice(device));

    PedoniModel* m = new PedoniModel();
    m->device = device;
    m->opt = *opt;
    m->size_x = size_x;
    m->size_y = size_y;
    m->rng.s = opt->seed;
    {
        const char* fs = std::getenv("PEDONI_FORCE_SIMPLE");
        m->force_simple = fs && fs[0] == '1';
#ifdef PEDONI_DIAGNOSTICS
        const char* ab = std::getenv("PEDONI_ABLATE");
        m->ablate = ab ? std::atoi(ab) : 0;
#endif
        const char* sg = std::getenv("PEDONI_SORT_GENERAL");
        m->sort_general = sg && sg[0] == '1';
        const char* nf = std::getenv("PEDONI_NO_FUSE_KEY");
        m->no_fuse_key = nf && nf[0] == '1';
        const char* nx = std::getenv("PEDONI_NO_XCD_REMAP");
        m->xcd_remap = !(nx && nx[0] == '1');
        const char* ft = nullptr;
#ifdef PEDONI_DIAGNOSTICS
        ft = std::getenv("PEDONI_FORCE_TRACE");
#endif
        if (ft && ft[0] == '1') {
            if (hipMalloc((void**)&m->d_trace, TRACE_WAVES * 8 * sizeof(unsigned long long)) != hipSuccess ||
                hipMemset(m->d_trace, 0, TRACE_WAVES * 8 * sizeof(unsigned long long)) != hipSuccess) {
                pedoni_hip_destroy(m);
                return fail(PEDONI_E_HIP, "create: trace buffer");
            }
        }
        const char* nto = std::getenv("PEDONI_NO_TILE_ORDER");
        m->tile_order_on = !(nto && nto[0] == '1');
        if (const char* gt = std::getenv("PEDONI_GRAPH_TICKS")) {
            const int k = std::atoi(gt);
            m->graph_long = k >= 16 ? 16u : (k >= 8 ? 8u : (k >= 4 ? 4u : 0u));
        }
        const char* ng = std::getenv("PEDONI_NO_GRAPH");
        m->use_graph = !(ng && ng[0] == '1');
        if (const char* fg = std::getenv("PEDONI_FORCE_GROUP")) {
            m->force_group = std::atoi(fg);
            if (m->force_group != 1 && m->force_group != 2 && m->force_group != 4) {
                pedoni_hip_destroy(m);
                return fail(PEDONI_E_INVALID, "create: PEDONI_FORCE_GROUP must be 1, 2 or 4");
            }
        }
        if (const char* fgs = std::getenv("PEDONI_FORCE_GROUP_SLOTS")) m->force_group_slots = std::atoi(fgs);
        if (const char* fk = std::getenv("PEDONI_FORCE_KERNEL")) {
            const std::string v(fk);
            const size_t colon = v.find(':');
            const std::string b = v.substr(0, colon);
            const int sl = colon == std::string::npos ? 0 : std::atoi(v.c_str() + colon + 1);
            const bool s94 = b == "s94", def = b == "default";
            const bool known = (s94 && (sl == 4 || sl == 5 || sl == 6 || sl == 8)) || (def && (sl == 5 || sl == 6 || sl == 8));
            if (!known) {
                pedoni_hip_destroy(m);
                return fail(PEDONI_E_INVALID, "create: PEDONI_FORCE_KERNEL must be s94:{4,5,6,8} or default:{5,6,8}");
            }
            m->force_choice = ForceChoice{s94 ? ForceBuild::S94 : ForceBuild::Default, sl};
        }
#ifdef PEDONI_DIAGNOSTICS
        const char* fp = std::getenv("PEDONI_FORCE_PERSIST");
        if (fp) m->force_persist = std::atoi(fp);      // 1 / 7, 6, 5: ticket form at that many waves per SIMD; 15-17: static strides
#endif
    }
    *out = nullptr;

    auto bail = [&](int rc) { pedoni_hip_destroy(m); return rc; };
#define C_TRY(expr) do { int rc2_ = (expr); if (rc2_ != PEDONI_OK) return bail(rc2_); } while (0)
#define C_HIP(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) return bail(fail(PEDONI_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e2_))); } while (0)

    C_HIP(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
    m->stream = m->own_stream;
    {
        // the interior rows of a split sharded tick run on a stream of the LOWEST priority: the edge rows'
        // small force launch, the pack and the exchange behind it -- the tick's critical path -- must get
        // the wave slots the interior's thousands of workgroups free, not queue behind them
        int least = 0, greatest = 0;
        C_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        C_HIP(hipStreamCreateWithPriority(&m->side_stream, hipStreamNonBlocking, least));
    }
    C_HIP(hipEventCreateWithFlags(&m->ev_sorted, hipEventDisableTiming));
    C_HIP(hipEventCreateWithFlags(&m->ev_interior, hipEventDisableTiming));

    // field maps: the texel rows [map_row_begin, map_row_end) of every map go to the device; the
    // pointers the kernels index are biased by -map_row_begin rows, so texel (y, x) stays at
    // base[y * cols + x] (never dereferenced outside the slice: device_math.hpp FieldView)
    const size_t slice_off = (size_t)map_row_begin * field_cols;
    const size_t texels = (size_t)(map_row_end - map_row_begin) * field_cols;
    C_TRY(dev_alloc(&m->d_distance_map, texels));
    C_HIP(hipMemcpy(m->d_distance_map, distance_map + slice_off, texels * sizeof(float), hipMemcpyHostToDevice));
    // (the potential maps in one allocation, map k at k * texels: device_math.hpp potential_map)
    C_TRY(dev_alloc(&m->d_pot, (size_t)n_maps * texels));
    for (uint32_t k = 0; k < n_maps; ++k) {
        if (!potential_maps[k]) return bail(fail(PEDONI_E_INVALID, "create: null potential map"));
        C_HIP(hipMemcpy(m->d_pot + (size_t)k * texels, potential_maps[k] + slice_off, texels * sizeof(float), hipMemcpyHostToDevice));
    }
    m->field.distance_map = (const float*)((uintptr_t)m->d_distance_map - slice_off * sizeof(float));
    m->field.pot_base = (const float*)((uintptr_t)m->d_pot - slice_off * sizeof(float));
    m->field.pot_stride = (int64_t)texels;
    m->field.y_lo = (int32_t)map_row_begin;
    m->field.y_hi = (int32_t)map_row_end;
    m->field.rows = (int32_t)field_rows;
    m->field.cols = (int32_t)field_cols;
    m->field.unit = field_unit;
    {
        int e = 0;
        const float mant = std::frexp(field_unit, &e);
        // 2^-100 .. 2^100: both unit and 1 / unit are normal floats
        m->field.unit_pow2 = (mant == 0.5f && e > -100 && e < 100) ? 1 : 0;
        m->field.inv_unit = m->field.unit_pow2 ? 1.0f / field_unit : 0.0f;
    }
    m->field.n_maps = n_maps;

    m->n_obstacles = n_obstacles;
    C_TRY(dev_alloc(&m->d_obstacles, n_obstacles));
    if (n_obstacles)
        C_HIP(hipMemcpy(m->d_obstacles, obstacles, n_obstacles * sizeof(PedoniObstacle), hipMemcpyHostToDevice));

    // neighbor grid, neighbor_grid.rs:14-20: shape = ceil(size / unit) as usize
    if (opt->use_neighbor_grid) {
        if (!(opt->neighbor_grid_unit > 0.0f))
            return bail(fail(PEDONI_E_INVALID, "create: neighbor_grid_unit must be > 0"));
        float fr = std::ceil(size_y / opt->neighbor_grid_unit);
        float fc = std::ceil(size_x / opt->neighbor_grid_unit);
        if (!(fr >= 1.0f) || !(fc >= 1.0f) || (double)fr * (double)fc > 1.0e9)
            return bail(fail(PEDONI_E_INVALID, "create: neighbor grid shape out of range"));
        m->grid.unit = opt->neighbor_grid_unit;
        m->grid.rows = (int32_t)fr;
        m->grid.cols = (int32_t)fc;
        m->n_cells = (uint32_t)m->grid.rows * (uint32_t)m->grid.cols;
        m->scan_cap = m->n_cells + 1;
        C_TRY(dev_alloc(&m->d_scan_in, m->scan_cap));
        C_HIP(hipMemset(m->d_scan_in, 0, (size_t)m->scan_cap * sizeof(uint32_t)));
        for (int k = 0; k < 2; ++k) {
            C_TRY(dev_alloc(&m->d_cs[k], m->scan_cap));
            C_HIP(hipMemset(m->d_cs[k], 0, (size_t)m->scan_cap * sizeof(uint32_t)));
        }
        C_TRY(dev_alloc(&m->d_row_count, (size_t)m->grid.rows + 1));
        C_HIP(hipMemset(m->d_row_count, 0, ((size_t)m->grid.rows + 1) * sizeof(uint32_t)));
        C_TRY(dev_alloc(&m->d_tickets, (size_t)9 * TICKET_STRIDE));     // + place_kernel's workgroups-done counter
        C_HIP(hipMemset(m->d_tickets, 0, (size_t)9 * TICKET_STRIDE * sizeof(uint32_t)));
        // per-cell early-out flags of the despawn test and the wall term (kernels.hpp): two launches over the
        // cells, once; PEDONI_NO_CELL_FLAGS=1 leaves them out (every agent samples: A/B runs, tests)
        const char* ncf = std::getenv("PEDONI_NO_CELL_FLAGS");
        if (!(ncf && ncf[0] == '1')) {
            uint32_t* own = nullptr;
            C_TRY(dev_alloc(&m->d_cell_flags, m->n_cells));
            C_TRY(dev_alloc(&own, m->n_cells));
            hipLaunchKernelGGL(cell_flags_own_kernel, dim3((m->n_cells + 255u) / 256u), dim3(256), 0, m->stream, m->field, m->grid, own);
            // (wall segments instead of the distance map: bit 31 from the obstacles themselves -- unless the
            // scenario has so many that the table would take seconds to build)
            if (!opt->use_distance_map)
                hipLaunchKernelGGL(cell_flags_segments_kernel, dim3((m->n_cells + 255u) / 256u), dim3(256), 0, m->stream, m->grid,
                                   m->d_obstacles, n_obstacles, (double)m->n_cells * (double)n_obstacles <= 4.0e9 ? 1 : 0, own);
            hipLaunchKernelGGL(cell_flags_block_kernel, dim3((m->n_cells + 255u) / 256u), dim3(256), 0, m->stream, m->grid, own,
                               m->d_cell_flags);
            const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(m->stream);
            hipFree(own);
            C_HIP(e1);
            C_HIP(e2);
        }
    }
    {
        int per_cu = 0, cus = 0;
        C_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, scan_rows_kernel<true>, SCAN_THREADS, 0));
        C_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        m->scan_wait_rows_max = (uint32_t)std::max(0, per_cu * cus / 2);
        if (const char* sw = std::getenv("PEDONI_SCAN_WAIT_ROWS_MAX")) m->scan_wait_rows_max = (uint32_t)std::atoi(sw);
    }
    m->band_lo = 0;
    m->band_hi = opt->use_neighbor_grid ? m->grid.rows : 0;

    C_TRY(dev_alloc(&m->d_live, 4));
    C_HIP(hipMemset(m->d_live, 0, 4 * sizeof(uint32_t)));
    m->field.status = m->d_live + 1;
    C_TRY(dev_alloc(&m->d_flags, 1));
    C_HIP(hipMemset(m->d_flags, 0, sizeof(SortFlags)));
    C_TRY(dev_alloc(&m->d_halo, 1));
    C_HIP(hipMemset(m->d_halo, 0, sizeof(HaloIn)));
    C_HIP(hipHostMalloc((void**)&m->h_pinned, 16 * sizeof(uint32_t), hipHostMallocDefault));
    C_TRY(ensure_capacity(m, std::max<uint32_t>(opt->initial_capacity, 1024)));
    C_HIP(hipDeviceSynchronize());
#undef C_TRY
#undef C_HIP
    *out = m;
    return PEDONI_OK;
}

void pedoni_hip_destroy(PedoniModel* m)
{
    if (!m) return;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    shard_detach_model(m);      // a shard that outlives its model fails with "null shard" from now on
    for (auto& p : m->ev_pool) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (int k = 0; k < 2; ++k) {
        hipFree(m->d_pos[k]); hipFree(m->d_velx[k]); hipFree(m->d_dest[k]);
    }
    hipFree(m->d_key); hipFree(m->d_slots);
    hipFree(m->d_tile_weight); hipFree(m->d_tile_order);
    hipFree(m->d_scan_in); hipFree(m->d_cs[0]); hipFree(m->d_cs[1]); hipFree(m->d_block_sums);
    hipFree(m->d_skey[0]); hipFree(m->d_skey[1]); hipFree(m->d_flags);
    hipFree(m->d_live); hipFree(m->d_acc); hipFree(m->d_halo);
    hipFree(m->d_spawners); hipFree(m->d_spawn_state);
    hipFree(m->d_row_count);
    hipFree(m->d_cell_flags);
    hipFree(m->d_tickets);
    hipFree(m->d_trace);
    if (m->h_pinned) hipHostFree(m->h_pinned);
    hipFree(m->d_distance_map);
    hipFree(m->d_pot);
    hipFree(m->d_obstacles);
    if (m->side_stream) { hipStreamSynchronize(m->side_stream); hipStreamDestroy(m->side_stream); }
    if (m->ev_sorted) hipEventDestroy(m->ev_sorted);
    if (m->ev_interior) hipEventDestroy(m->ev_interior);
    for (hipEvent_t e : m->ev_tick) if (e) hipEventDestroy(e);
    for (auto& g : m->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    for (auto& lg : m->long_graphs) for (auto& g : lg) if (g.exec) hipGraphExecDestroy(g.exec);
    if (m->own_stream) hipStreamDestroy(m->own_stream);
    delete m;
}

int pedoni_hip_append(PedoniModel* m, const float* pos_xy, const uint32_t* destination,
                      const float* desired_speed, const float* vel_xy, uint32_t n)
{
    TRY(bind(m));
    return append(m, pos_xy, destination, desired_speed, vel_xy, n);
}

int pedoni_hip_sort_despawn(PedoniModel* m)
{
    TRY(bind(m));
    return sort_despawn(m);
}

int pedoni_hip_spawn_pedestrians(PedoniModel* m, const PedoniPedestrian* peds, uint32_t n)
{
    TRY(bind(m));
    if (n) {
        if (!peds) return fail(PEDONI_E_INVALID, "spawn_pedestrians: null peds");
        std::vector<float> pos(2 * (size_t)n);
        std::vector<uint32_t> dest(n);
        for (uint32_t i = 0; i < n; ++i) {
            pos[2 * i] = peds[i].x;
            pos[2 * i + 1] = peds[i].y;
            // sfm.rs:52 `p.destination as u32`
            dest[i] = (uint32_t)peds[i].destination;
        }
        TRY(append(m, pos.data(), dest.data(), nullptr, nullptr, n));
    }
    return sort_despawn(m);
}

int pedoni_hip_update_states(PedoniModel* m)
{
    TRY(bind(m));
    return update_states(m);
}

namespace {
// lib.rs:70-84 + sfm.rs:49-56 on the device: append this tick's Poisson arrivals
int device_spawn(PedoniModel* m)
{
    if (m->gap_end != m->n_upper)
        return fail(PEDONI_E_INVALID, "device spawning: host-appended agents are pending; run a pass first");
    // same bound bookkeeping as halo_unpack: re-read the device's count every 16 ticks
    if (++m->ticks_since_tighten >= 16 || (uint64_t)m->n_upper + m->spawn_cap > m->cap) {
        uint32_t live = 0;
        TRY(sync_live_count(m, &live));
        m->ticks_since_tighten = 0;
    }
    TRY(ensure_capacity(m, m->n_upper + m->spawn_cap));
    Timed t(m, PEDONI_K_OTHER);
    if (t.rc) return t.rc;
    hipLaunchKernelGGL(spawn_kernel, dim3(1), dim3(64), 0, m->stream, m->d_spawners, m->n_spawners,
                       m->d_spawn_state, m->n_upper, m->spawn_cap, m->d_pos[m->pv], m->d_velx[m->pv],
                       m->d_dest[m->vd], m->d_halo);
    HIP_TRY(hipGetLastError());
    m->gap_end = m->n_upper;        // the spawned agents start here
    m->n_upper += m->spawn_cap;     // host bound; the device knows the true count
    m->sorted = false;
    return PEDONI_OK;
}
} // namespace

namespace {
// Steady state (no agents appended, no exchange, no device spawning, keys fused by the last
// update): a tick is scan -> place -> reorder -> force with every branch decided on the
// device, and after TWO ticks every ping-pong index of the host is back where it was.  That
// pair is captured once as a hipGraph and replayed: one graph launch instead of eight kernel
// launches, which is what bounds the small configurations (C2: 1e5 agents) -- the kernels are
// the same, so the results are too.  Anything that changes a baked-in argument (host bound,
// buffers, stream, options) invalidates the capture.
bool graphable(const PedoniModel* m)
{
    return m->use_graph && m->stream != nullptr && m->opt.use_neighbor_grid && !m->force_simple && !m->sort_general &&
           !m->no_fuse_key && !m->n_spawners && !m->halo_cap && m->have_old &&
           m->keys_valid && !m->sorted && m->gap_end == m->n_upper && m->n_upper > m->base &&
           m->grid.rows <= 0xffff && m->grid.cols <= 0xffff;
}

// Records `ticks` ticks from the current host state; with `keep` the recording becomes that parity's
// replayable pair, without it only the host bookkeeping advances (nothing runs on the device).
int capture_ticks(PedoniModel* m, int ticks, PedoniModel::TickGraph* keep)
{
    const uint32_t n_upper = m->n_upper, base = m->base, parity = m->tick_parity & 1u;
    const int pv = m->pv, vd = m->vd, cs = m->cs, sk = m->sk;
    HIP_TRY(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    int rc = PEDONI_OK;
    for (int t = 0; t < ticks && rc == PEDONI_OK; ++t) {
        rc = sort_despawn(m);
        if (rc == PEDONI_OK) rc = update_states(m);
    }
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(m->stream, &graph);
    if (rc != PEDONI_OK) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(PEDONI_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    if (!keep) { hipGraphDestroy(graph); return PEDONI_OK; }
    e = hipGraphInstantiate(&keep->exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { keep->exec = nullptr; return fail(PEDONI_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    // the capture ran the host bookkeeping of two ticks: the indices are back at the key
    if (m->pv != pv || m->vd != vd || m->cs != cs || m->sk != sk || (m->tick_parity & 1u) != parity ||
        m->n_upper != n_upper)
        return fail(PEDONI_E_HIP, "tick graph: host state did not return after two ticks");
    keep->n_upper = n_upper; keep->base = base; keep->pv = pv; keep->vd = vd; keep->cs = cs; keep->sk = sk;
    keep->stream = m->stream;
    keep->valid = true;
    return PEDONI_OK;
}

bool graph_matches(const PedoniModel* m, const PedoniModel::TickGraph& g)
{
    return g.valid && g.exec && g.n_upper == m->n_upper && g.base == m->base && g.pv == m->pv && g.vd == m->vd &&
           g.cs == m->cs && g.sk == m->sk && g.stream == m->stream;
}

// the host bookkeeping a tick advances (sort_despawn, update_states): a capture runs it without running
// anything on the device, so a capture that fails midway must put it back
struct HostBook {
    int pv, vd, cs, sk;
    uint32_t tick_parity, n_upper, gap_end, ticks_since_tighten;
    bool have_old, keys_valid, counts_dirty, sorted, halo_keys_done, tickets_fresh;
    static HostBook of(const PedoniModel* m)
    {
        return HostBook{m->pv, m->vd, m->cs, m->sk, m->tick_parity, m->n_upper, m->gap_end, m->ticks_since_tighten,
                        m->have_old, m->keys_valid, m->counts_dirty, m->sorted, m->halo_keys_done, m->tickets_fresh};
    }
    void restore(PedoniModel* m) const
    {
        m->pv = pv; m->vd = vd; m->cs = cs; m->sk = sk; m->tick_parity = tick_parity; m->n_upper = n_upper;
        m->gap_end = gap_end; m->ticks_since_tighten = ticks_since_tighten; m->have_old = have_old;
        m->keys_valid = keys_valid; m->counts_dirty = counts_dirty; m->sorted = sorted;
        m->halo_keys_done = halo_keys_done; m->tickets_fresh = tickets_fresh;
    }
};

int tick_graph_pair(PedoniModel* m)
{
    const uint32_t parity = m->tick_parity & 1u;
    PedoniModel::TickGraph& g = m->graphs[parity];
    if (!graph_matches(m, g)) {
        // Up to four captures follow (this pair; one discarded tick, the other pair, one discarded tick), each
        // advancing the host bookkeeping by what it records.  If any of them fails, nothing has run on the
        // device: the bookkeeping goes back to where it stood, the graphs are dropped and this model ticks
        // eagerly from now on (it used to return with the host up to three ticks ahead of the device).
        const HostBook before = HostBook::of(m);
        auto captured = [&]() -> int {
            if (g.exec) { hipGraphExecDestroy(g.exec); g.exec = nullptr; }
            g.valid = false;
            TRY(capture_ticks(m, 2, &g));
            // the pair that starts one tick later, while we are at it: a caller that mixes single ticks
            // in (every n-th tick event-timed) then never meets a capture + instantiate in mid-run
            PedoniModel::TickGraph& o = m->graphs[parity ^ 1u];
            if (!(o.valid && o.exec && o.n_upper == m->n_upper && o.base == m->base && o.stream == m->stream)) {
                if (o.exec) { hipGraphExecDestroy(o.exec); o.exec = nullptr; }
                o.valid = false;
                TRY(capture_ticks(m, 1, nullptr));
                TRY(capture_ticks(m, 2, &o));
                TRY(capture_ticks(m, 1, nullptr));
                if ((m->tick_parity & 1u) != parity || !graph_matches(m, g))
                    return fail(PEDONI_E_HIP, "tick graph: host state did not return after the second capture");
            }
            return PEDONI_OK;
        };
        const int rc = captured();
        if (rc != PEDONI_OK) {
            before.restore(m);
            m->drop_graphs();
            m->use_graph = false;
            return rc;
        }
    }
    {
        Range replay("tick pair (captured graph replay)");
        HIP_TRY(hipGraphLaunch(g.exec, m->stream));
    }
    // host state after a pair of ticks == before it (see above); the flags the eager path
    // would have left: keys fused, counts pending, order not sorted
    m->keys_valid = true;
    m->counts_dirty = true;
    m->sorted = false;
    return PEDONI_OK;
}

// a run of 16 / 8 / 4 ticks (which = 0 / 1 / 2; an even number: the host bookkeeping is back where it was) in one
// graph launch; captured per parity the first time it is wanted
constexpr uint32_t LONG_RUNS[3] = {16u, 8u, 4u};
int tick_graph_long(PedoniModel* m, int which)
{
    const uint32_t parity = m->tick_parity & 1u;
    PedoniModel::TickGraph& g = m->long_graphs[which][parity];
    if (!graph_matches(m, g)) {
        // this run's graph and, while we are at it, the one that starts on the other half of the ping-pong buffers
        // (one discarded tick, the capture, one discarded tick -- as for the pairs): a caller that mixes single
        // event-timed ticks in meets no capture + instantiate in mid-run (bench.py: all of them happen while
        // the crowd settles, none inside the timed region).  A capture that fails puts the bookkeeping back.
        const HostBook before = HostBook::of(m);
        auto captured = [&]() -> int {
            if (g.exec) { hipGraphExecDestroy(g.exec); g.exec = nullptr; }
            g.valid = false;
            TRY(capture_ticks(m, (int)LONG_RUNS[which], &g));
            PedoniModel::TickGraph& o = m->long_graphs[which][parity ^ 1u];
            if (!(o.valid && o.exec && o.n_upper == m->n_upper && o.base == m->base && o.stream == m->stream)) {
                if (o.exec) { hipGraphExecDestroy(o.exec); o.exec = nullptr; }
                o.valid = false;
                TRY(capture_ticks(m, 1, nullptr));
                TRY(capture_ticks(m, (int)LONG_RUNS[which], &o));
                TRY(capture_ticks(m, 1, nullptr));
                if ((m->tick_parity & 1u) != parity || !graph_matches(m, g))
                    return fail(PEDONI_E_HIP, "tick graph: host state did not return after the second capture");
            }
            return PEDONI_OK;
        };
        const int rc = captured();
        if (rc != PEDONI_OK) {
            before.restore(m);
            m->drop_graphs();
            m->use_graph = false;
            return rc;
        }
    }
    {
        Range replay("run of ticks (captured graph replay)");
        HIP_TRY(hipGraphLaunch(g.exec, m->stream));
    }
    m->keys_valid = true;
    m->counts_dirty = true;
    m->sorted = false;
    return PEDONI_OK;
}
} // namespace

int pedoni_hip_tick_n(PedoniModel* m, uint32_t steps)
{
    TRY(bind(m));
    // with profiling on, the kernels of every profile_every-th tick are event-timed (those
    // ticks launch eagerly); all other ticks may replay the captured pair
    auto sampled = [&](uint64_t t) { return m->sampled(t); };
    uint32_t s = 0;
    int rc = PEDONI_OK;
    while (s < steps && rc == PEDONI_OK) {
        if (m->graph_long && steps - s >= 4 && graphable(m)) {
            // how many ticks ahead are plain ones (not event-timed, inside this call)?
            uint32_t run = 0;
            while (run < m->graph_long && run < steps - s && !sampled(m->tick_counter + run)) ++run;
            int which = -1;
            for (int k = 0; k < 3 && which < 0; ++k)
                if (LONG_RUNS[k] <= run) which = k;
            if (which >= 0) {
                m->profile_now = false;
                rc = tick_graph_long(m, which);
                s += LONG_RUNS[which];
                m->tick_counter += LONG_RUNS[which];
                continue;
            }
        }
        if (steps - s >= 2 && !sampled(m->tick_counter) && !sampled(m->tick_counter + 1) && graphable(m)) {
            m->profile_now = false;
            rc = tick_graph_pair(m);
            s += 2;
            m->tick_counter += 2;
            continue;
        }
        m->profile_now = sampled(m->tick_counter);
        if (m->n_spawners) rc = device_spawn(m);
        if (rc == PEDONI_OK) rc = sort_despawn(m);
        if (rc == PEDONI_OK) rc = update_states(m);
        s += 1;
        m->tick_counter += 1;
    }
    m->profile_now = true;
    return rc;
}

int pedoni_hip_set_spawners(PedoniModel* m, const PedoniSpawner* spawners, uint32_t n,
                            uint64_t position_rng_state, uint32_t max_per_tick)
{
    TRY(bind(m));
    if (n && !spawners) return fail(PEDONI_E_INVALID, "set_spawners: null spawners");
    if (m->halo_cap) return fail(PEDONI_E_INVALID, "set_spawners: not supported for a band of a sharded run");
    if (n && max_per_tick == 0) return fail(PEDONI_E_INVALID, "set_spawners: max_per_tick must be > 0");
    if (n && !m->opt.use_neighbor_grid)
        return fail(PEDONI_E_INVALID, "set_spawners: device spawning needs the neighbor grid "
                                      "(the brute-force option path spawns on the host)");
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->drop_graphs();
    if (m->n_spawners) { // hand the desired-speed stream back to the host side
        SpawnState st{};
        HIP_TRY(hipMemcpy(&st, m->d_spawn_state, sizeof st, hipMemcpyDeviceToHost));
        m->rng.s = st.rng_v0;
    }
    hipFree(m->d_spawners);
    m->d_spawners = nullptr;
    m->n_spawners = 0;
    m->spawn_cap = 0;
    if (n == 0) return PEDONI_OK;
    std::vector<SpawnerDev> dev(n);
    for (uint32_t k = 0; k < n; ++k) {
        if (!(spawners[k].frequency >= 0.0)) return fail(PEDONI_E_INVALID, "set_spawners: bad frequency");
        dev[k] = SpawnerDev{spawners[k].x0, spawners[k].y0, spawners[k].x1, spawners[k].y1,
                            spawners[k].destination, 0u, std::exp(-(spawners[k].frequency / 10.0))};
    }
    TRY(dev_alloc(&m->d_spawners, n));
    HIP_TRY(hipMemcpy(m->d_spawners, dev.data(), n * sizeof(SpawnerDev), hipMemcpyHostToDevice));
    if (!m->d_spawn_state) TRY(dev_alloc(&m->d_spawn_state, 1));
    SpawnState st{position_rng_state, m->rng.s};
    HIP_TRY(hipMemcpy(m->d_spawn_state, &st, sizeof st, hipMemcpyHostToDevice));
    m->n_spawners = n;
    m->spawn_cap = max_per_tick;
    return PEDONI_OK;
}

int pedoni_hip_get_spawn_rng(PedoniModel* m, uint64_t* position_rng_state, uint64_t* speed_rng_state)
{
    TRY(bind(m));
    SpawnState st{0, m->rng.s};
    if (m->n_spawners) {
        HIP_TRY(hipMemcpyAsync(&st, m->d_spawn_state, sizeof st, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    if (position_rng_state) *position_rng_state = st.rng_pos;
    if (speed_rng_state) *speed_rng_state = st.rng_v0;
    return PEDONI_OK;
}

int pedoni_hip_set_speed_rng(PedoniModel* m, uint64_t speed_rng_state)
{
    TRY(bind(m));
    if (m->n_spawners)
        return fail(PEDONI_E_INVALID, "set_speed_rng: uninstall the device spawners first "
                                      "(they own the stream while installed)");
    m->rng.s = speed_rng_state;
    return PEDONI_OK;
}

int pedoni_hip_tick(PedoniModel* m, PedoniStepMetrics* metrics)
{
    TRY(bind(m));
    using clk = std::chrono::steady_clock;
    if (!m->ev_tick[0]) {
        HIP_TRY(hipEventCreate(&m->ev_tick[0]));
        if (hipEventCreate(&m->ev_tick[1]) != hipSuccess) {
            hipEventDestroy(m->ev_tick[0]);
            m->ev_tick[0] = nullptr;
            return fail(PEDONI_E_HIP, "hipEventCreate failed");
        }
    }
    HIP_TRY(hipStreamSynchronize(m->stream));
    auto t0 = clk::now();
    TRY(sort_despawn(m));
    HIP_TRY(hipStreamSynchronize(m->stream));
    auto t1 = clk::now();
    HIP_TRY(hipEventRecord(m->ev_tick[0], m->stream));
    TRY(update_states(m));
    HIP_TRY(hipEventRecord(m->ev_tick[1], m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    auto t2 = clk::now();
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, m->ev_tick[0], m->ev_tick[1]) != hipSuccess) ms = -1.0f;
    if (metrics) {
        uint32_t live = 0;
        TRY(sync_live_count(m, &live));
        metrics->active_ped_count = (int32_t)live;                            // lib.rs:95
        metrics->time_spawn = std::chrono::duration<double>(t1 - t0).count(); // lib.rs:86
        metrics->time_calc_state = std::chrono::duration<double>(t2 - t1).count(); // lib.rs:91
        metrics->time_calc_state_kernel = ms >= 0 ? ms * 1e-3 : -1.0;         // lib.rs:98 (None upstream)
    }
    return PEDONI_OK;
}

int pedoni_hip_get_pedestrian_count(PedoniModel* m, int32_t* count)
{
    TRY(bind(m));
    if (!count) return fail(PEDONI_E_INVALID, "null count");
    uint32_t live = 0;
    TRY(sync_live_count(m, &live));
    // agents appended by the host since the last pass are part of `self.pedestrians` upstream
    // too (device-stored ranges are bounds, not counts: they are consumed by the next pass)
    const bool device_ranges = m->halo_cap || m->n_spawners;
    *count = (int32_t)(live + (device_ranges ? 0u : m->n_upper - m->gap_end));
    return PEDONI_OK;
}

int pedoni_hip_download(PedoniModel* m, float* pos_xy, uint32_t* destination, float* vel_xy,
                        float* desired_speed, uint32_t cap, uint32_t* n)
{
    TRY(bind(m));
    uint32_t live = 0;
    TRY(sync_live_count(m, &live));
    uint32_t appended = m->n_upper - m->gap_end;
    uint32_t total = live + appended;
    if (n) *n = total;
    // live agents [0, live) then appended agents [gap_end, n_upper)
    uint32_t n1 = std::min(live, cap), n2 = std::min(appended, cap - n1);
    auto copy = [&](void* dst, const void* src, size_t elem) -> int {
        if (!dst) return PEDONI_OK;
        if (n1)
            HIP_TRY(hipMemcpyAsync(dst, (const char*)src + (size_t)m->base * elem, n1 * elem,
                                   hipMemcpyDeviceToHost, m->stream));
        if (n2)
            HIP_TRY(hipMemcpyAsync((char*)dst + (size_t)n1 * elem,
                                   (const char*)src + (size_t)m->gap_end * elem, n2 * elem,
                                   hipMemcpyDeviceToHost, m->stream));
        return PEDONI_OK;
    };
    TRY(copy(pos_xy, m->d_pos[m->pv], sizeof(float2)));
    std::vector<float4> velx;
    if (vel_xy || desired_speed) {
        velx.resize((size_t)n1 + n2);
        TRY(copy(velx.data(), m->d_velx[m->pv], sizeof(float4)));
    }
    TRY(copy(destination, m->d_dest[m->vd], sizeof(uint32_t)));
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (size_t i = 0; i < velx.size(); ++i) {          // {vx, vy, vl, desired speed} -> the two host arrays
        if (vel_xy) { vel_xy[2 * i] = velx[i].x; vel_xy[2 * i + 1] = velx[i].y; }
        if (desired_speed) desired_speed[i] = velx[i].w;
    }
    return PEDONI_OK;
}

int pedoni_hip_list_pedestrians(PedoniModel* m, PedoniPedestrian* out, uint32_t cap, uint32_t* n)
{
    TRY(bind(m));
    uint32_t total = 0;
    TRY(pedoni_hip_download(m, nullptr, nullptr, nullptr, nullptr, 0, &total));
    if (n) *n = total;
    uint32_t k = std::min(total, cap);
    if (!out || k == 0) return PEDONI_OK;
    std::vector<float> pos(2 * (size_t)k);
    std::vector<uint32_t> dest(k);
    uint32_t got = 0;
    TRY(pedoni_hip_download(m, pos.data(), dest.data(), nullptr, nullptr, k, &got));
    for (uint32_t i = 0; i < k; ++i) {
        out[i].x = pos[2 * i];
        out[i].y = pos[2 * i + 1];
        out[i].destination = dest[i];
    }
    return PEDONI_OK;
}

int pedoni_hip_clear(PedoniModel* m)
{
    TRY(bind(m));
    HIP_TRY(hipMemcpyAsync(m->d_live, &m->base, sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->opt.use_neighbor_grid) {
        std::vector<uint32_t> fill((size_t)m->n_cells + 1, m->base);
        HIP_TRY(hipMemcpy(m->d_cs[m->cs], fill.data(), fill.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    m->n_upper = m->gap_end = m->base;
    m->have_old = false;
    m->keys_valid = false;
    m->sorted = false;
    m->drop_graphs();       // the captured tick pair bakes in band, bounds and buffers
    return PEDONI_OK;
}

int pedoni_hip_neighbor_grid_indices(PedoniModel* m, uint32_t* out, uint32_t cap, uint32_t* len)
{
    TRY(bind(m));
    if (!m->opt.use_neighbor_grid) { if (len) *len = 0; return PEDONI_OK; }
    uint32_t n = m->n_cells + 1;
    if (len) *len = n;
    if (out && cap) {
        HIP_TRY(hipMemcpyAsync(out, m->d_cs[m->cs], (size_t)std::min(n, cap) * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    return PEDONI_OK;
}

int pedoni_hip_neighbor_grid_shape(PedoniModel* m, uint32_t* rows, uint32_t* cols)
{
    if (!m) return fail(PEDONI_E_INVALID, "null model");
    if (rows) *rows = m->opt.use_neighbor_grid ? (uint32_t)m->grid.rows : 0;
    if (cols) *cols = m->opt.use_neighbor_grid ? (uint32_t)m->grid.cols : 0;
    return PEDONI_OK;
}

int pedoni_hip_cell_flags(PedoniModel* m, uint32_t* out, uint32_t cap, uint32_t* len)
{
    TRY(bind(m));
    if (!len) return fail(PEDONI_E_INVALID, "cell_flags: null len");
    *len = m->d_cell_flags ? m->n_cells : 0u;
    if (!out || !m->d_cell_flags) return PEDONI_OK;
    if (cap < m->n_cells) return fail(PEDONI_E_CAPACITY, "cell_flags: buffer too small");
    HIP_TRY(hipMemcpy(out, m->d_cell_flags, (size_t)m->n_cells * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PEDONI_OK;
}

int pedoni_hip_tile_order(PedoniModel* m, uint32_t* order, uint32_t* tile_weight, uint32_t cap, uint32_t* n_blocks)
{
    TRY(bind(m));
    if (!n_blocks) return fail(PEDONI_E_INVALID, "tile_order: null n_blocks");
    HIP_TRY(hipStreamSynchronize(m->stream));
    *n_blocks = m->tile_order_blocks;
    if (!m->tile_order_blocks || (!order && !tile_weight)) return PEDONI_OK;
    if (cap < m->tile_order_blocks) return fail(PEDONI_E_CAPACITY, "tile_order: buffer too small");
    if (order) HIP_TRY(hipMemcpy(order, m->d_tile_order, (size_t)m->tile_order_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (tile_weight) {
        std::vector<uint32_t> w((size_t)m->tile_order_blocks * FORCE_WAVES);
        HIP_TRY(hipMemcpy(w.data(), m->d_tile_weight, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t t = 0; t < m->tile_order_blocks; ++t) {
            tile_weight[t] = 0;
            for (int k = 0; k < FORCE_WAVES; ++k) tile_weight[t] += w[(size_t)t * FORCE_WAVES + k];
        }
    }
    return PEDONI_OK;
}

int pedoni_hip_calc_accelerations(PedoniModel* m, float* acc_xy, uint32_t cap)
{
    TRY(bind(m));
    if (!m->sorted) return fail(PEDONI_E_INVALID, "calc_accelerations needs a sorted state");
    if (!acc_xy) return fail(PEDONI_E_INVALID, "null acc");
    if (m->n_upper == m->base) return PEDONI_OK;
    if (m->acc_cap < m->n_upper) {
        hipFree(m->d_acc);
        m->d_acc = nullptr;
        TRY(dev_alloc(&m->d_acc, m->cap));
        m->acc_cap = m->cap;
    }
    TRY(launch_force(m, m->d_acc));
    uint32_t live = 0;
    TRY(sync_live_count(m, &live));
    uint32_t k = std::min(live, cap);
    if (k) HIP_TRY(hipMemcpy(acc_xy, m->d_acc + m->base, (size_t)k * sizeof(float2), hipMemcpyDeviceToHost));
    return PEDONI_OK;
}

int pedoni_hip_set_stream(PedoniModel* m, void* hip_stream, int32_t use_library_stream)
{
    TRY(bind(m));
    TRY(drain_events(m));
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->stream = use_library_stream ? m->own_stream : (hipStream_t)hip_stream;
    m->drop_graphs();
    return PEDONI_OK;
}

int pedoni_hip_get_stream(PedoniModel* m, void** hip_stream)
{
    if (!m || !hip_stream) return fail(PEDONI_E_INVALID, "null argument");
    *hip_stream = (void*)m->stream;
    return PEDONI_OK;
}

int pedoni_hip_synchronize(PedoniModel* m)
{
    TRY(bind(m));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return PEDONI_OK;
}

int pedoni_hip_profile(PedoniModel* m, int32_t enable)
{
    TRY(bind(m));
    TRY(drain_events(m));
    m->profile_mask = (uint32_t)enable & ((1u << PEDONI_N_KERNELS) - 1u);
    // the event pairs of the first timed ticks exist before those ticks run (creating them between
    // the launches of a tick puts the host behind the device)
    while (m->profile_mask && m->ev_pool.size() < 256) {
        EventPair p{};
        HIP_TRY(hipEventCreate(&p.a));
        if (hipEventCreate(&p.b) != hipSuccess) { hipEventDestroy(p.a); return fail(PEDONI_E_HIP, "hipEventCreate failed"); }
        m->ev_pool.push_back(p);
    }
    return PEDONI_OK;
}

int pedoni_hip_profile_every(PedoniModel* m, uint32_t every_ticks)
{
    TRY(bind(m));
    if (every_ticks == 0) return fail(PEDONI_E_INVALID, "profile_every: must be >= 1");
    m->profile_every = every_ticks;
    m->profile_burst = 1;
    m->profile_phase = 0;
    return PEDONI_OK;
}

int pedoni_hip_profile_burst(PedoniModel* m, uint32_t every_ticks, uint32_t burst_ticks)
{
    TRY(bind(m));
    if (every_ticks == 0 || burst_ticks == 0 || burst_ticks > every_ticks)
        return fail(PEDONI_E_INVALID, "profile_burst: need 1 <= burst <= every");
    m->profile_every = every_ticks;
    m->profile_burst = burst_ticks;
    m->profile_phase = m->tick_counter;     // the next tick starts a burst
    return PEDONI_OK;
}

int pedoni_hip_kernel_times(PedoniModel* m, PedoniKernelTimes* out, int32_t reset)
{
    TRY(bind(m));
    TRY(drain_events(m));
    if (out) *out = m->times;
    if (reset) m->times = PedoniKernelTimes{};
    return PEDONI_OK;
}

// ---- row-band sharding ------------------------------------------------------------------------
int pedoni_hip_set_band(PedoniModel* m, int32_t row_begin, int32_t row_end, uint32_t halo_cap)
{
    TRY(bind(m));
    if (!m->opt.use_neighbor_grid)
        return fail(PEDONI_E_INVALID, "set_band: sharding needs the neighbor grid");
    if (row_begin < 0 || row_end > m->grid.rows || row_begin >= row_end)
        return fail(PEDONI_E_INVALID, "set_band: bad row range");
    if (m->n_upper != m->base)
        return fail(PEDONI_E_INVALID, "set_band: the model must hold no agents");
    if (halo_cap > (1u << 24)) return fail(PEDONI_E_INVALID, "set_band: halo capacity too large");
    m->band_lo = row_begin;
    m->band_hi = row_end;
    m->halo_cap = halo_cap;
    m->base = halo_cap;
    m->drop_graphs();       // (band_lo / band_hi are arguments of the captured kernels)
    TRY(ensure_capacity(m, m->base + std::max<uint32_t>(m->opt.initial_capacity, 1024)));
    return pedoni_hip_clear(m);
}

int pedoni_hip_halo_bytes(uint32_t cap_each, uint64_t* bytes)
{
    if (!bytes) return fail(PEDONI_E_INVALID, "null out");
    *bytes = 2ull * (PEDONI_HALO_HEADER_WORDS + (uint64_t)cap_each * PEDONI_HALO_RECORD_WORDS) * 4ull;
    return PEDONI_OK;
}

namespace {
// `updated` = read positions / velocities from the buffer update_states is writing (the
// boundary rows of a split tick are already there; the buffers flip when the tick ends)
int halo_pack_from(PedoniModel* m, void* send_dev, uint32_t cap_each, bool updated, hipStream_t on = nullptr)
{
    const int src = updated ? 1 - m->pv : m->pv;
    Timed t(m, on ? -1 : PEDONI_K_HALO_PACK);     // (a launch on another stream is not event-timed)
    if (t.rc) return t.rc;
    hipLaunchKernelGGL(halo_pack_kernel, dim3(2), dim3(256), 0, on ? on : m->stream, m->d_pos[src],
                       m->d_velx[src], m->d_dest[m->vd], m->d_cs[m->cs], m->grid,
                       m->band_lo, m->band_hi, cap_each, (uint32_t*)send_dev);
    HIP_TRY(hipGetLastError());
    return PEDONI_OK;
}
} // namespace

int pedoni_hip_halo_pack(PedoniModel* m, void* send_dev, uint32_t cap_each)
{
    TRY(bind(m));
    if (!send_dev) return fail(PEDONI_E_INVALID, "halo_pack: null buffer");
    if (cap_each != m->halo_cap || cap_each == 0)
        return fail(PEDONI_E_INVALID, "halo_pack: cap_each differs from set_band's halo capacity");
    if (!m->have_old)
        return fail(PEDONI_E_INVALID, "halo_pack: needs a sorted order (run sort_despawn once after loading)");
    return halo_pack_from(m, send_dev, cap_each, /*updated=*/false);
}

namespace {
// how many sharded ticks may pass before the host re-reads the device's live count (one stream sync)
uint32_t tighten_every()
{
    static const uint32_t v = [] { const char* e = std::getenv("PEDONI_TIGHTEN_EVERY"); return e ? (uint32_t)std::max(1, std::atoi(e)) : 8u; }();
    return v;
}
} // namespace

int halo_unpack_on(PedoniModel* m, const void* from_below_dev, const void* from_above_dev, uint32_t cap_each,
                   hipStream_t on);

int pedoni_hip_halo_unpack(PedoniModel* m, const void* from_below_dev, const void* from_above_dev,
                           uint32_t cap_each)
{
    return halo_unpack_on(m, from_below_dev, from_above_dev, cap_each, nullptr);
}

// Would the next unpack re-read the live count (and pull the host's bound of the arrays back to it)?
// Then it must not run beside the force launch of the tick before, whose surplus threads are still
// writing DEAD keys to the slots between the live count and the OLD bound -- where the list from
// above would land.
bool halo_unpack_would_tighten(const PedoniModel* m, uint32_t cap_each, bool from_below, bool from_above)
{
    const uint32_t grow = cap_each * std::max(1u, (from_below ? 1u : 0u) + (from_above ? 1u : 0u));
    return m->ticks_since_tighten + 1u >= tighten_every() || (uint64_t)m->n_upper + grow > m->cap;
}

// `on` = null: the model's stream.  Another stream (the shard's communication stream, right behind
// the exchange that filled the lists): the kernel runs while the force launch of the tick before is
// still at work -- the lists land OUTSIDE that launch's agents ([base - n, base) and behind the
// host's bound of the stored agents), the keys they get are slots that launch does not write, and
// the cell / row counts both add to are integer atomics.
int halo_unpack_on(PedoniModel* m, const void* from_below_dev, const void* from_above_dev, uint32_t cap_each,
                   hipStream_t on)
{
    TRY(bind(m));
    if (cap_each != m->halo_cap || cap_each == 0)
        return fail(PEDONI_E_INVALID, "halo_unpack: cap_each differs from set_band's halo capacity");
    // agents appended by the host since the last pass would count as neither own nor received
    // (the above list starts at gap_end): they must go through a pass first
    if (m->gap_end != m->n_upper)
        return fail(PEDONI_E_INVALID, "halo_unpack: host-appended agents are pending; run sort_despawn after append");
    // The received lists land outside the own agents ([base - n, base) and behind everything
    // stored), and after the sort every live agent -- own, from below, from above -- sits in
    // [base, live): the host's bound of the end of the arrays must grow by the capacity of
    // EVERY list received (at least one: the slots behind the stored agents are keyed even
    // with no band above; all `grow` of them get a key, DEAD beyond the list from above),
    // although the true count (on the device) barely changes.  Re-read the count every
    // 8 ticks -- one stream sync per 8 ticks -- so launches never cover more than ~7 % idle
    // threads; and always before the arrays would have to grow.
    const uint32_t grow = cap_each * std::max(1u, (from_below_dev ? 1u : 0u) + (from_above_dev ? 1u : 0u));
    if (++m->ticks_since_tighten >= tighten_every() || (uint64_t)m->n_upper + grow > m->cap) {
        uint32_t live = 0;
        TRY(sync_live_count(m, &live));
        m->ticks_since_tighten = 0;
    }
    TRY(ensure_capacity(m, m->n_upper + grow));
    const uint32_t words_each = PEDONI_HALO_HEADER_WORDS + cap_each * PEDONI_HALO_RECORD_WORDS;
    // a rank's buffer is [down list][up list]: the band below sends us its UP list, the band
    // above its DOWN list
    const uint32_t* below = from_below_dev ? (const uint32_t*)from_below_dev + words_each : nullptr;
    const uint32_t* above = (const uint32_t*)from_above_dev;
    Timed t(m, on ? -1 : PEDONI_K_HALO_UNPACK);      // (only launches on the model's stream are event-timed)
    if (t.rc) return t.rc;
    hipLaunchKernelGGL(halo_unpack_kernel, dim3(blocks_for(cap_each + grow, 256)), dim3(256), 0,
                       on ? on : m->stream, below, above, cap_each, grow, m->base, m->n_upper, m->d_pos[m->pv],
                       m->d_velx[m->pv], m->d_dest[m->vd], m->d_halo, m->field, m->grid,
                       m->band_lo, m->band_hi, m->tick_parity & 1u, m->d_flags, m->d_key, m->d_scan_in,
                       m->d_row_count);
    HIP_TRY(hipGetLastError());
    m->counts_dirty = true;
    m->gap_end = m->n_upper;      // the above list starts here
    m->n_upper += grow;           // host bound; the device knows the true count
    m->halo_keys_done = true;     // the exchanged agents already carry their keys
    m->sorted = false;
    return PEDONI_OK;
}

int pedoni_hip_halo_tick(PedoniModel* m, const void* from_below_dev, const void* from_above_dev,
                         void* send_dev, uint32_t cap_each)
{
    if (!send_dev) return fail(PEDONI_E_INVALID, "halo_tick: null send buffer");
    TRY(pedoni_hip_halo_unpack(m, from_below_dev, from_above_dev, cap_each));
    TRY(sort_despawn(m));
    TRY(update_states(m));
    return pedoni_hip_halo_pack(m, send_dev, cap_each);
}

// The same tick in two halves, so that the exchange of the NEXT tick's lists overlaps the
// bulk of this tick's force kernel:
//   begin: unpack, sort/despawn, force + integrate the rows next to the band's edges, pack
//          (the caller now starts the all-gather of `send_dev`, asynchronously)
//   end:   force + integrate the interior rows
int pedoni_hip_halo_tick_begin(PedoniModel* m, const void* from_below_dev, const void* from_above_dev,
                               void* send_dev, uint32_t cap_each)
{
    if (!send_dev) return fail(PEDONI_E_INVALID, "halo_tick_begin: null send buffer");
    TRY(pedoni_hip_halo_unpack(m, from_below_dev, from_above_dev, cap_each));
    TRY(sort_despawn(m));
    if (m->band_hi - m->band_lo < 6 || m->force_simple) {
        // band too thin to split: whole tick now, nothing left for _end
        TRY(update_states(m));
        m->split_pending = false;
        return halo_pack_from(m, send_dev, cap_each, false);
    }
    // interior rows on the side stream, concurrently with the (small, latency-bound) edge-row
    // launch, the pack and the exchange the caller starts next
    // (edge rows are enqueued first so that their few blocks are dispatched ahead of the
    // interior's thousands)
    HIP_TRY(hipEventRecord(m->ev_sorted, m->stream));
    TRY(launch_force(m, nullptr, /*part=*/1));
    HIP_TRY(hipStreamWaitEvent(m->side_stream, m->ev_sorted, 0));
    TRY(launch_force(m, nullptr, /*part=*/2, m->side_stream));
    HIP_TRY(hipEventRecord(m->ev_interior, m->side_stream));
    m->split_pending = true;
    return halo_pack_from(m, send_dev, cap_each, /*updated=*/true);
}

int pedoni_hip_halo_tick_end(PedoniModel* m)
{
    TRY(bind(m));
    if (!m->split_pending) return PEDONI_OK;
    HIP_TRY(hipStreamWaitEvent(m->stream, m->ev_interior, 0)); // join before the next pass
    after_update(m);
    m->split_pending = false;
    return PEDONI_OK;
}

int pedoni_hip_force_kernel_info(PedoniModel* m, uint32_t n_agents, char* name, uint32_t name_cap,
                                 uint32_t* agents_per_wave)
{
    if (!m) return fail(PEDONI_E_INVALID, "null model");
    const bool fast = m->opt.math_mode == PEDONI_MATH_FAST;
    std::string sym;
    uint32_t per_wave = 64;
    if (!m->opt.use_neighbor_grid || m->force_simple) {
        sym = std::string("force_kernel_simple<") + (fast ? "1" : "0") + ">";
    } else {
        const ForcePlan c = plan_force(m, n_agents, true);
        const std::string mode = fast ? "1" : "0";
        if (c.group > 1) {
            sym = "force_kernel_queue_group<" + mode + ", " + std::to_string(c.slots) + ", " + std::to_string(c.group) + ">";
            per_wave = 64u / (uint32_t)c.group;
        } else {
            sym = std::string(c.build == ForceBuild::S94 ? "force_kernel_queue_s94<" : "force_kernel_queue<") + mode + ", " +
                  std::to_string(c.slots) + ">";
        }
    }
    if (name && name_cap) {
        std::strncpy(name, sym.c_str(), name_cap - 1);
        name[name_cap - 1] = '\0';
    }
    if (agents_per_wave) *agents_per_wave = per_wave;
    return PEDONI_OK;
}

int pedoni_hip_owned_count(PedoniModel* m, int32_t* count)
{
    TRY(bind(m));
    if (!count) return fail(PEDONI_E_INVALID, "null count");
    if (!m->opt.use_neighbor_grid || (!m->sorted && !m->have_old))
        return pedoni_hip_get_pedestrian_count(m, count);
    uint32_t lo = 0, hi = 0;
    HaloIn h{};
    HIP_TRY(hipMemcpyAsync(m->h_pinned + 2, m->d_live + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(&lo, m->d_cs[m->cs] + (size_t)m->band_lo * m->grid.cols, sizeof(uint32_t),
                           hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(&hi, m->d_cs[m->cs] + (size_t)m->band_hi * m->grid.cols, sizeof(uint32_t),
                           hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(&h, m->d_halo, sizeof(HaloIn), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    TRY(check_status(m->h_pinned[2]));
    if (h.error & 1u) return fail(PEDONI_E_CAPACITY, "halo list overflow: raise the halo capacity");
    if (h.error & 2u)
        return fail(PEDONI_E_INVALID, "an agent left its band by more than one grid row in one tick");
    if (h.error & 4u)
        return fail(PEDONI_E_CAPACITY, "boundary rows hold more agents than 8 x halo capacity: raise it");
    if (h.error & 8u)
        return fail(PEDONI_E_CAPACITY, "a tick spawned more agents than max_per_tick");
    *count = (int32_t)(hi - lo);
    return PEDONI_OK;
}

#ifdef PEDONI_DIAGNOSTICS
int pedoni_hip_debug_force_trace(PedoniModel* m, uint64_t* sums7, int32_t reset)
{
    TRY(bind(m));
    if (!m->d_trace) return fail(PEDONI_E_INVALID, "force trace: create the model with PEDONI_FORCE_TRACE=1");
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (sums7) {
        std::vector<unsigned long long> rec(TRACE_WAVES * 8);
        HIP_TRY(hipMemcpy(rec.data(), m->d_trace, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int k = 0; k < 7; ++k) sums7[k] = 0;
        for (size_t w = 0; w < TRACE_WAVES; ++w)
            for (int k = 0; k < 7; ++k) sums7[k] += rec[8 * w + k];
    }
    if (reset) HIP_TRY(hipMemset(m->d_trace, 0, TRACE_WAVES * 8 * sizeof(unsigned long long)));
    return PEDONI_OK;
}

int pedoni_hip_debug_set_status(PedoniModel* m, uint32_t status_word)
{
    TRY(bind(m));
    HIP_TRY(hipMemcpyAsync(m->d_live + 1, &status_word, sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return PEDONI_OK;
}

// diagnostics: the raw per-wave trace records (8 words each; see force_queue_tile_group TRACE)
int pedoni_hip_debug_force_trace_raw(PedoniModel* m, uint64_t* out, uint32_t n_waves)
{
    TRY(bind(m));
    if (!m->d_trace || !out || n_waves > TRACE_WAVES) return fail(PEDONI_E_INVALID, "force trace raw: bad arguments");
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(out, m->d_trace, (size_t)n_waves * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PEDONI_OK;
}

// diagnostics: the PEDONI_ABLATE bit mask of the force kernel (timing only, results wrong), settable
// between ticks so that ONE launch of a warmed-up crowd can be timed with parts switched off
int pedoni_hip_debug_set_ablate(PedoniModel* m, uint32_t bits)
{
    TRY(bind(m));
    m->ablate = (int)(bits & 0xffu);
    m->place_ablate = bits >> 8;                          // bits 8 and up: place_kernel's switches (kernels_diag.hpp SwitchDiag)
    m->drop_graphs();
    return PEDONI_OK;
}
#endif // PEDONI_DIAGNOSTICS

// ---- device math self-test -----------------------------------------------------------------
} // extern "C"

namespace {
template <int MODE>
__global__ void selftest_kernel(int op, const float* a, const float* b, float* out, uint32_t n)
{
    __shared__ uint64_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = EXP2F_TAB[threadIdx.x];
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
    case 0: r = fdiv<MODE>(a[i], b[i]); break;
    case 1: r = fsqrt<MODE>(a[i]); break;
    case 2: r = fexp<MODE>(a[i], tab); break;
    case 3: r = div_03<MODE>(a[i]); break;
    case 4: r = div_02<MODE>(a[i]); break;
    case 5: r = __int_as_float(f32_as_i32(a[i])); break;   // raw i32 bits
    case 6: r = div_core(a[i], recip_refined(b[i])); break;
    default: r = 0.0f;
    }
    out[i] = r;
}
template <int MODE>
__global__ void selftest_pair_kernel(const float2* pos, const float2* e, const float2* pos_i,
                                     const float2* vel_i, float2* acc, uint32_t n)
{
    __shared__ uint64_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = EXP2F_TAB[threadIdx.x];
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v2 a = mk(acc[i].x, acc[i].y);
    pair_force<MODE>(mk(pos[i].x, pos[i].y), mk(e[i].x, e[i].y), mk(pos_i[i].x, pos_i[i].y),
                     mk(vel_i[i].x, vel_i[i].y), a, tab);
    acc[i] = make_float2(a.x, a.y);
}
__global__ void selftest_field_kernel(const float* grid, int32_t rows, int32_t cols, const float* px,
                                      const float* py, float2* grad, float* centre, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c;
    MapDims md;
    md.rows = rows; md.cols = cols; md.y_lo = 0; md.y_hi = rows; md.status = nullptr;
    v2 g = sobel_fast(grid, md, px[i], py[i], &c);
    grad[i] = make_float2(g.x, g.y);
    centre[i] = c;
}
} // namespace

extern "C" int pedoni_hip_selftest_field(int device, const float* grid, uint32_t rows, uint32_t cols,
                                         const float* px, const float* py, float* grad_xy,
                                         float* centre, uint32_t n)
{
    if (!grid || !px || !py || !grad_xy || !centre || rows == 0 || cols == 0 ||
        rows > 0x7fffffffu || cols > 0x7fffffffu)
        return fail(PEDONI_E_INVALID, "selftest_field: bad argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(PEDONI_E_NO_DEVICE, "selftest: no HIP device");
    HIP_TRY(hipSetDevice(device));
    if (n == 0) return PEDONI_OK;
    float *dg = nullptr, *dx = nullptr, *dy = nullptr, *dc = nullptr;
    float2* dgrad = nullptr;
    const size_t cells = (size_t)rows * cols;
    int rc = PEDONI_OK;
    if (hipMalloc((void**)&dg, cells * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&dx, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&dy, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&dc, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&dgrad, n * sizeof(float2)) != hipSuccess ||
        hipMemcpy(dg, grid, cells * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dx, px, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dy, py, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PEDONI_E_HIP, "selftest_field: device allocation / copy failed");
    if (rc == PEDONI_OK) {
        hipLaunchKernelGGL(selftest_field_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dg, (int32_t)rows,
                           (int32_t)cols, dx, dy, dgrad, dc, n);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpy(grad_xy, dgrad, n * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(centre, dc, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PEDONI_E_HIP, "selftest_field: launch / copy failed");
    }
    hipFree(dg); hipFree(dx); hipFree(dy); hipFree(dc); hipFree(dgrad);
    return rc;
}

extern "C" int pedoni_hip_selftest_pair(int device, int32_t math_mode, const float* pos_xy,
                                        const float* e_xy, const float* pos_i_xy,
                                        const float* vel_i_xy, float* acc_xy, uint32_t n)
{
    if (!pos_xy || !e_xy || !pos_i_xy || !vel_i_xy || !acc_xy)
        return fail(PEDONI_E_INVALID, "selftest_pair: null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(PEDONI_E_NO_DEVICE, "selftest: no HIP device");
    HIP_TRY(hipSetDevice(device));
    if (n == 0) return PEDONI_OK;
    float2* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const float* h[5] = {pos_xy, e_xy, pos_i_xy, vel_i_xy, acc_xy};
    int rc = PEDONI_OK;
    for (int k = 0; k < 5 && rc == PEDONI_OK; ++k) {
        if (hipMalloc((void**)&d[k], (size_t)n * sizeof(float2)) != hipSuccess ||
            hipMemcpy(d[k], h[k], (size_t)n * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(PEDONI_E_HIP, "selftest_pair: device allocation / copy failed");
    }
    if (rc == PEDONI_OK) {
        if (math_mode == PEDONI_MATH_FAST)
            hipLaunchKernelGGL(selftest_pair_kernel<1>, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], n);
        else
            hipLaunchKernelGGL(selftest_pair_kernel<0>, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], n);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpy(acc_xy, d[4], (size_t)n * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PEDONI_E_HIP, "selftest_pair: launch / copy failed");
    }
    for (int k = 0; k < 5; ++k) hipFree(d[k]);
    return rc;
}

extern "C" int pedoni_hip_selftest_math(int device, int32_t op, int32_t math_mode, const float* a,
                                        const float* b, float* out, uint32_t n)
{
    if (!a || !out || ((op == 0 || op == 6) && !b) || op < 0 || op > 6)
        return fail(PEDONI_E_INVALID, "selftest: bad arguments");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(PEDONI_E_NO_DEVICE, "selftest: no HIP device");
    HIP_TRY(hipSetDevice(device));
    if (n == 0) return PEDONI_OK;
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = PEDONI_OK;
    if (hipMalloc((void**)&da, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&db, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&dout, n * sizeof(float)) != hipSuccess ||
        hipMemcpy(da, a, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        (b && hipMemcpy(db, b, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess))
        rc = fail(PEDONI_E_HIP, "selftest_math: device allocation / copy failed");
    if (rc == PEDONI_OK) {
        if (math_mode == PEDONI_MATH_FAST)
            hipLaunchKernelGGL(selftest_kernel<1>, dim3((n + 255) / 256), dim3(256), 0, 0, op, da, db, dout, n);
        else
            hipLaunchKernelGGL(selftest_kernel<0>, dim3((n + 255) / 256), dim3(256), 0, 0, op, da, db, dout, n);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PEDONI_E_HIP, "selftest_math: launch / copy failed");
    }
    hipFree(da); hipFree(db); hipFree(dout);
    return rc;
}

// ---- multi-GPU driver (pedoni_shard_*) ---------------------------------------------------------
#include "shard.hpp"

// ---- opt-in GPU field builder (pedoni_hip_eikonal) -----------------------------------------------
#include "eikonal.hpp"
