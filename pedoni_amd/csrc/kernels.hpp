// kernels.hpp -- gfx950 kernels of the per-timestep pedestrian update.
//
// Device-resident counterpart of SocialForceModel::spawn_pedestrians (sort/despawn half,
// pedoni-simulator/src/models/sfm.rs:58-88) and ::update_states (sfm.rs:91-255).
//
// Data layout in HBM (structure of arrays, all fp32 / u32):
//   pos[2][cap] float2, velx[2][cap] float4 {vx, vy, vl, desired_speed}   ping-pong twice per
//   tick (sort, integrate); vl = |v| * 0.1 of sfm.rs:144, filled in by the sort pass (neighbour_vl)
//   destination[2][cap] u32   ping-pong once per tick (sort)
//   key[cap] u32 (next cell id or DEAD), slots[cap] u32 (general-form scratch)
//   skey[2][cap] u32 packed (cy << 16 | cx) cell of each sorted agent (unfused K_KEY only)
//   cell_count[cells+1], cell_start[2][cells+1] u32 (= the reference's neighbor_grid_indices;
//   ping-pong: the gather sort form reads last tick's while writing this tick's)
//   field maps: distance_map + n potential maps, row-major (y, x) f32
//   row_count[rows+1] u32: members per grid row (the scan's top level)
// Kernels per tick in steady state: scan -> place (-> reorder: no-op) -> force; the per-cell
// and per-row counts the scan consumes are accumulated by the force kernel's tail (integer
// atomics).
#pragma once

#include "block_order.hpp"
#include "device_math.hpp"
#include "pedoni_hip.h"

namespace pedoni {

using PedoniObstacleDev = ::PedoniObstacle;

constexpr uint32_t DEAD = 0xffffffffu;
constexpr uint32_t TICKET_STRIDE = 32;   // words: one 128-byte line per XCD's ticket word (force_kernel_queue_persist)

// Diagnostics hooks.  The tile functions of the force kernel and the body of the place kernel take a policy
// object: the product kernels are instantiated with NoDiag -- every hook an empty inline, every switch a
// constant false, nothing of it in their code -- and csrc/kernels_diag.hpp (diagnostics build only) has the
// ones that time a wave's phases or switch parts of a kernel off for ablation runs.
struct ForceArgs;
struct NoDiag {
    __device__ __forceinline__ NoDiag() {}
    __device__ __forceinline__ NoDiag(const ForceArgs&, uint32_t /*wave record*/) {}
    __device__ __forceinline__ void lap(int /*phase*/) {}          // the phase that just ended
    __device__ __forceinline__ void flush() {}                     // the wave's lanes leave the tile
    __device__ __forceinline__ constexpr bool off(uint32_t /*part*/) const { return false; }
};

struct GridView {
    float unit;
    int32_t rows, cols; // NeighborGrid.shape = (rows, cols) (neighbor_grid.rs:14-20)
};

// neighbor_grid.rs:27-29: (pos / unit).as_ivec2() then Index::index_checked; false = the
// agent is outside the grid and is never binned
__device__ __forceinline__ bool cell_xy(const GridView& g, v2 pos, int32_t& cx, int32_t& cy)
{
    cx = f32_as_i32(pos.x / g.unit);
    cy = f32_as_i32(pos.y / g.unit);
    return !(cx < 0 || cy < 0 || cy >= g.rows || cx >= g.cols);
}

// field.rs:235-239 get_potential(dest, pos) > 0.25 (sfm.rs:69,82); a destination with no
// map panics upstream (index out of bounds) -- here the agent is dropped instead.
__device__ __forceinline__ bool survives(const FieldView& f, v2 pos, uint32_t dest)
{
    if (dest >= f.n_maps) return false;
    v2 q = field_coord(f, pos);
    return bilinear(potential_map(f, dest), dims_of(f), q.x, q.y) > 0.25f;
}

// ---- per-cell early-out flags (built once, at pedoni_hip_create) --------------------------------
// Two terms of the tick cost texel gathers whose RESULT is known in advance for most agents:
//   * the despawn test of the next pass (sfm.rs:69 `get_potential(dest, pos) > 0.25`): util::bilinear
//     is a combination of 4 texels with non-negative weights that add up to 1 within a few ulp
//     (util.rs:47-56; out-of-shape texels read 1e12), so it is > 0.25 for certain when every texel a
//     position in the agent's NEXT neighbor-grid cell can touch is >= 0.26 (4 % of margin against
//     ~1e-6 of rounding);
//   * the wall term (sfm.rs:188-192) `direction * (2 * exp(-distance / 0.2))`: f32::exp(-d / 0.2) is
//     exactly 0 for d > 20.8 m (its result would lie below half the smallest denormal), so the term is
//     (+-0, +-0) whenever `direction` is finite, i.e. the Sobel gradient is finite and not (0, 0) -- and
//     acc + (+-0) == acc bit for bit unless acc is itself +-0 (a case the kernel sends down the full path).
// One word per neighbor-grid cell c, indexed by the cell of the agent's CURRENT position:
//   bit m (m < 31): potential map m is >= 0.26 on every texel that util::bilinear can read for a
//           position inside the 3 x 3 cells around c (an agent that ends its step outside that
//           block, or with a NaN position, takes the sampled test);
//   bit 31 (use_distance_map; the explicit-segment form: cell_flags_segments_kernel below): on every texel
//           the distance map's 3 x 3 Sobel taps + centre can read for a position inside c, the map lies in [21, 4096] AND steps by at least CELL_FLAG_STEP from texel to texel, with
//           one sign, along x or along y -- then the left and right (or upper and lower) tap columns
//           differ by >= 8 steps in exact arithmetic against < 0.02 of accumulated fp32 rounding: the
//           gradient cannot vanish, and every tap is finite.
// A texel outside the field reads 1e12 (util.rs:53-56): fine for the despawn bits, fatal for bit 31; a
// texel outside the rows a band has uploaded clears both.  Everything here errs towards a cleared bit:
// a cleared bit only means the kernel samples as before.
constexpr uint32_t CELL_FLAG_WALL = 0x80000000u;
constexpr uint32_t CELL_FLAG_MAPS = 31u;
constexpr float CELL_FLAG_POTENTIAL_MIN = 0.26f;
constexpr float CELL_FLAG_WALL_MIN = 21.0f, CELL_FLAG_WALL_MAX = 4096.0f, CELL_FLAG_STEP = 0.03f;

// texel index range [lo, hi] (inclusive) that the bilinear taps of a position in grid cell `c` of width
// `gu` can read, with one texel of slack either side for the roundings of pos / unit - 0.5 and of the
// cell's own pos / gu (neighbor_grid.rs:27: `as i32` truncates towards zero, so cell 0 also holds (-gu, 0))
__device__ __forceinline__ void cell_texel_range(int32_t c, double gu, double fu, int32_t apron, int32_t& lo, int32_t& hi)
{
    double x_lo = c == 0 ? -gu : (double)c * gu, x_hi = ((double)c + 1.0) * gu;
    x_lo -= 1e-6 * (x_lo < 0 ? -x_lo : x_lo) + 1e-6;
    x_hi += 1e-6 * x_hi + 1e-6;
    const double q_lo = __builtin_floor(x_lo / fu - 0.5), q_hi = __builtin_floor(x_hi / fu - 0.5);
    // (clamped far outside any map: the loops below only ever compare these with the map's shape)
    lo = (int32_t)__builtin_fmax(q_lo - 1.0 - (double)apron, -1.0e9);
    hi = (int32_t)__builtin_fmin(q_hi + 2.0 + (double)apron, 1.0e9);
}

// pass 1: the flags of each cell's OWN positions
__global__ void cell_flags_own_kernel(FieldView f, GridView g, uint32_t* __restrict__ own)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (uint32_t)g.rows * (uint32_t)g.cols) return;
    const int32_t cy = (int32_t)(c / (uint32_t)g.cols), cx = (int32_t)(c - (uint32_t)cy * (uint32_t)g.cols);
    int32_t x0, x1, y0, y1;
    cell_texel_range(cx, (double)g.unit, (double)f.unit, 0, x0, x1);
    cell_texel_range(cy, (double)g.unit, (double)f.unit, 0, y0, y1);
    uint32_t bits = 0;
    // (a neighbor grid much coarser than the field: not worth a table)
    if ((int64_t)(x1 - x0 + 3) * (int64_t)(y1 - y0 + 3) > 4096) { own[c] = 0; return; }
    const uint32_t n_maps = f.n_maps < CELL_FLAG_MAPS ? f.n_maps : CELL_FLAG_MAPS;
    for (uint32_t m = 0; m < n_maps; ++m) {
        MapPtr map = as_map(potential_map(f, m));
        bool ok = true;
        for (int32_t y = y0; y <= y1 && ok; ++y) {
            if (y < 0 || y >= f.rows) continue;                       // out of shape: reads 1e12
            if (y < f.y_lo || y >= f.y_hi) { ok = false; break; }     // not in this band's slice
            for (int32_t x = x0; x <= x1; ++x) {
                if (x < 0 || x >= f.cols) continue;
                if (!(map[(int64_t)y * f.cols + x] >= CELL_FLAG_POTENTIAL_MIN)) { ok = false; break; }   // (NaN clears)
            }
        }
        if (ok) bits |= 1u << m;
    }
    {
        // the Sobel taps reach one texel further each way
        const int32_t wx0 = x0 - 1, wx1 = x1 + 1, wy0 = y0 - 1, wy1 = y1 + 1;
        MapPtr map = as_map(f.distance_map);
        bool ok = wx0 >= 0 && wy0 >= f.y_lo && wy0 >= 0 && wx1 < f.cols && wy1 < f.y_hi && wy1 < f.rows;
        bool up_x = true, down_x = true, up_y = true, down_y = true;
        for (int32_t y = wy0; ok && y <= wy1; ++y) {
            for (int32_t x = wx0; x <= wx1; ++x) {
                const float v = map[(int64_t)y * f.cols + x];
                if (!(v >= CELL_FLAG_WALL_MIN && v <= CELL_FLAG_WALL_MAX)) { ok = false; break; }
                if (x < wx1) {
                    const float d = map[(int64_t)y * f.cols + x + 1] - v;
                    up_x = up_x && d >= CELL_FLAG_STEP; down_x = down_x && d <= -CELL_FLAG_STEP;
                }
                if (y < wy1) {
                    const float d = map[(int64_t)(y + 1) * f.cols + x] - v;
                    up_y = up_y && d >= CELL_FLAG_STEP; down_y = down_y && d <= -CELL_FLAG_STEP;
                }
            }
        }
        if (ok && (up_x || down_x || up_y || down_y)) bits |= CELL_FLAG_WALL;
    }
    own[c] = bits;
}

// bit 31 when the wall term comes from the explicit segments (use_distance_map = false, sfm.rs:193-236): every
// obstacle adds normalize(nearest) * (2 * exp(-min_d / 0.2)) -- (+-0, +-0) once min_d > 20.8 m, and an obstacle
// the agent is inside of adds nothing at all (:211-217).  The four segments of an obstacle (:200-209) all lie
// within |width| / 2 of its centre line, so min_d >= distance(position, centre line) - |width| / 2: the bit is
// set when that bound exceeds 21 m for EVERY obstacle from anywhere in the cell (distance from the cell's
// centre minus its half diagonal).  A NaN in an obstacle clears it.  Replaces the distance-map bit 31 of
// `own` (the two are never both in use).
__global__ void cell_flags_segments_kernel(GridView g, const PedoniObstacle* __restrict__ obs, uint32_t n_obs, int32_t enable,
                                           uint32_t* __restrict__ own)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (uint32_t)g.rows * (uint32_t)g.cols) return;
    if (!enable) { own[c] &= ~CELL_FLAG_WALL; return; }        // (no claim about the segments: only the distance map's bit goes)
    const int32_t cy = (int32_t)(c / (uint32_t)g.cols), cx = (int32_t)(c - (uint32_t)cy * (uint32_t)g.cols);
    const double gu = (double)g.unit;
    // the cell's extent in positions (cell 0 also holds (-gu, 0): `as i32` truncates towards zero), a little padded
    const double x0 = (cx == 0 ? -gu : cx * gu) - 1e-3, x1 = (cx + 1.0) * gu + 1e-3;
    const double y0 = (cy == 0 ? -gu : cy * gu) - 1e-3, y1 = (cy + 1.0) * gu + 1e-3;
    const double mx = 0.5 * (x0 + x1), my = 0.5 * (y0 + y1);
    const double reach = 0.5 * __builtin_sqrt((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0));
    bool far = true;
    for (uint32_t o = 0; o < n_obs && far; ++o) {
        const double ax = obs[o].x0, ay = obs[o].y0, bx = (double)obs[o].x1 - ax, by = (double)obs[o].y1 - ay;
        const double px = mx - ax, py = my - ay, bb = bx * bx + by * by;
        double t = bb > 0.0 ? (px * bx + py * by) / bb : 0.0;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double dx = px - t * bx, dy = py - t * by;
        const double w = obs[o].width;
        const double bound = __builtin_sqrt(dx * dx + dy * dy) - reach - 0.5 * (w < 0.0 ? -w : w);
        far = bound > (double)CELL_FLAG_WALL_MIN;              // (false for NaN)
    }
    own[c] = (own[c] & ~CELL_FLAG_WALL) | (far ? CELL_FLAG_WALL : 0u);
}

// pass 2: a despawn bit holds for the whole 3 x 3 block the agent can end its step in
__global__ void cell_flags_block_kernel(GridView g, const uint32_t* __restrict__ own, uint32_t* __restrict__ flags)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (uint32_t)g.rows * (uint32_t)g.cols) return;
    const int32_t cy = (int32_t)(c / (uint32_t)g.cols), cx = (int32_t)(c - (uint32_t)cy * (uint32_t)g.cols);
    uint32_t all = ~CELL_FLAG_WALL;
    for (int32_t y = max(cy - 1, 0); y <= min(cy + 1, g.rows - 1); ++y)
        for (int32_t x = max(cx - 1, 0); x <= min(cx + 1, g.cols - 1); ++x) all &= own[(int64_t)y * g.cols + x];
    flags[c] = all | (own[c] & CELL_FLAG_WALL);
}

// ---- the sort/despawn pass (sfm.rs:58-77) on the device -----------------------------------
// The reference bins every agent, then walks the cells row-major and each cell's list in
// insertion order: a STABLE sort by cell id, fused with the despawn test.  Two device forms
// produce exactly that order:
//
//  gather form (steady state).  Agents are still in last tick's sorted order and move
//    far less than one cell per tick, so the members of new cell c all sit in the 3 x 3
//    OLD cells around c -- three contiguous index ranges.  One thread per cell counts /
//    copies the agents of those ranges whose new key is c, in index order: no atomics, no
//    ranks, and the scan of the ranges is itself the reference's insertion order.
//  general form.  Any agent that is new (appended) or moved farther than one cell raises
//    a device flag in K_KEY; then the same launches take their other branch: per-cell
//    counts by integer atomics (exact whatever the arrival order), a provisional slot per
//    agent, and K_REORDER restores insertion order by counting the cell-mates with a
//    smaller previous index.
// The branch is chosen on the device, per tick, with no host round trip.

struct SortFlags {
    uint32_t far[2]; // far[tick & 1] != 0 -> general form this tick
};

__device__ __forceinline__ uint32_t pack_cell(uint32_t cx, uint32_t cy) { return (cy << 16) | cx; }

// Sticky device status word (PedoniModel::d_live[1]); every host read of the live count
// returns PEDONI_E_HIP / PEDONI_E_CAPACITY while a bit is set -- nothing continues silently.
constexpr uint32_t STATUS_SCAN_MISMATCH = 1u; // a row's cell counts do not add up to its row count
constexpr uint32_t STATUS_LIVE_OVERFLOW = 2u; // more live agents than the host's bound of the arrays
constexpr uint32_t STATUS_EDGE_WAIT = 8u;     // edge_wait_kernel gave up: the edge-first force launch never signalled

// Every stored key is followed by one count on its cell AND one on its grid row: the scan
// (scan_rows_kernel) turns the row totals into each row's first index without any
// inter-workgroup hand-off.  Integer atomics: exact in any arrival order.  The row add is
// aggregated over the wave first (sorted agents: a wave spans one or two rows), so it costs
// one or two atomics per wave.  Call with the lanes that have a key to count; `todo` false
// lanes only take part in the ballots.
__device__ __forceinline__ void count_key(uint32_t* __restrict__ cell_count, uint32_t* __restrict__ row_count,
                                          bool todo, uint32_t k, uint32_t cy)
{
    // (one atomic per agent.  Adding a run of equal keys -- neighbouring lanes, the agents being in
    // cell order -- by its first lane, 0.44 atomics per agent, was built and verified: no faster)
    if (todo) atomicAdd(&cell_count[k], 1u);
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        const unsigned long long pending = __ballot(todo);
        if (pending == 0ull) break;
        const int leader = __ffsll((long long)pending) - 1;
        const uint32_t row = (uint32_t)__shfl((int)cy, leader, 64);
        const bool same = todo && cy == row;
        const unsigned long long group = __ballot(same);
        if ((int)lane == leader) atomicAdd(&row_count[row], (uint32_t)__popcll(group));
        todo = todo && !same;
    }
}

// ---- K_KEY (PEDONI_K_BIN) ----------------------------------------------------------------
// One thread per stored agent.  Slots [live, gap_end) hold agents despawned by earlier
// ticks (the host only knows an upper bound of the live count) and are skipped.
struct HaloIn {
    // the first three words describe agents stored by the device since the last pass and are
    // cleared by the pass that consumes them
    uint32_t n_below; // agents received from the band below: stored at [base - n_below, base)
    uint32_t n_above; // agents received from the band above / spawned on the device: stored at
                      // [gap_end, gap_end + n_above)
    uint32_t counted; // 1: the appended range's length is n_above (not the host's bound)
    uint32_t error;   // sticky: bit 0 = a sender overflowed its list, bit 1 = an agent left its band
                      // by > 1 row, bit 2 = edge rows exceed their launch, bit 3 = spawn overflow
    uint32_t sharded; // 1 once the model exchanges halos
};

__global__ void key_kernel(const float2* __restrict__ pos, const uint32_t* __restrict__ dest,
                           uint32_t i0, uint32_t n_total, uint32_t base,
                           const uint32_t* __restrict__ live_count, uint32_t gap_end,
                           const HaloIn* __restrict__ halo, FieldView field, GridView grid,
                           int32_t band_lo, int32_t band_hi, const uint32_t* __restrict__ skey_old,
                           int32_t force_general, uint32_t parity, SortFlags* __restrict__ flags,
                           uint32_t* __restrict__ key, uint32_t* __restrict__ cell_count,
                           uint32_t* __restrict__ row_count)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && force_general) atomicOr(&flags->far[parity], 1u);
    uint32_t i = i0 + t;
    if (i >= n_total) return;
    uint32_t live = *live_count;
    uint32_t n_below = halo->n_below;
    uint32_t app_end = (halo->sharded || halo->counted) ? gap_end + halo->n_above : n_total;
    bool own = i >= base && i < live;
    bool received = (i < base && i >= base - n_below) || (i >= gap_end && i < app_end);
    uint32_t k = DEAD;
    int32_t cx = 0, cy = 0;
    if (own || received) {
        float2 p = pos[i];
        v2 pp = mk(p.x, p.y);
        // sharded runs keep only the band's rows plus one ghost row either side (tested before
        // the potential is sampled: a band may hold only its own rows of the maps)
        if (cell_xy(grid, pp, cx, cy) && cy >= band_lo - 1 && cy <= band_hi && survives(field, pp, dest[i])) {
            {
                k = (uint32_t)cy * (uint32_t)grid.cols + (uint32_t)cx;
                // agents appended since the last pass force the general form -- except the
                // exchanged lists of a sharded run, which can only land in the four boundary
                // rows, where count/write/reorder use the general form anyway
                bool boundary = halo->sharded && (cy <= band_lo || cy >= band_hi - 1);
                bool far = received && !boundary;
                if (!received && !force_general) {
                    uint32_t old = skey_old[i];
                    int32_t ox = (int32_t)(old & 0xffffu), oy = (int32_t)(old >> 16);
                    far = abs(cx - ox) > 1 || abs(cy - oy) > 1;
                }
                if (far) atomicOr(&flags->far[parity], 1u);
            }
        }
    }
    key[i] = k;
    count_key(cell_count, row_count, k != DEAD, k, (uint32_t)cy);
}

// the three old index ranges that can hold members of new cell (cx, cy)
struct CellRanges { uint32_t lo[3], hi[3]; };
__device__ __forceinline__ CellRanges old_ranges(const uint32_t* __restrict__ cs_old, const GridView& g,
                                                 int32_t cx, int32_t cy)
{
    CellRanges r;
    int32_t x0 = max(cx - 1, 0), x1 = min(cx + 1, g.cols - 1);
    // all six words requested at once (a row outside the grid reads row cy's words instead and is emptied
    // afterwards): a branch per row made three dependent round trips of them
    uint32_t lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t y = cy - 1 + k;
        const int64_t off = (int64_t)((y < 0 || y >= g.rows) ? cy : y) * g.cols;
        lo[k] = cs_old[off + x0];
        hi[k] = cs_old[off + x1 + 1];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t y = cy - 1 + k;
        const bool in = !(y < 0 || y >= g.rows);
        r.lo[k] = in ? lo[k] : 0u;
        r.hi[k] = in ? hi[k] : 0u;
    }
    return r;
}

// ---- which cells sort in which form ----------------------------------------------------------
// In a sharded run the cells of rows <= lo and >= hi-1 (ghost rows and the owned rows next
// to them) also receive the exchanged lists, which sit outside the old ranges: those cells
// always take the general form; all other rows keep the gather form.
struct BandView { int32_t lo, hi, sharded; };
__device__ __forceinline__ bool general_cell(const SortFlags* flags, uint32_t parity, const BandView& b,
                                             int32_t cy)
{
    return flags->far[parity] != 0 || (b.sharded && (cy <= b.lo || cy >= b.hi - 1));
}

// no-grid variant (sfm.rs:78-88): survivors keep their order; key = 1/0 flag to be scanned
__global__ void flag_kernel(const float2* __restrict__ pos, const uint32_t* __restrict__ dest,
                            uint32_t n_total, const uint32_t* __restrict__ live_count,
                            uint32_t gap_end, FieldView field, uint32_t* __restrict__ flag)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    uint32_t live = *live_count;
    uint32_t a = 0;
    if (i < live || i >= gap_end) {
        float2 p = pos[i];
        a = survives(field, mk(p.x, p.y), dest[i]) ? 1u : 0u;
    }
    flag[i] = a;
}

// ---- K_SCAN: exclusive prefix sum, 2048 elements per 256-thread block -----------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_PER_THREAD;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, sets total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave_sums,
                                                         uint32_t& total)
{
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) lds_wave_sums[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
    int n_waves = blockDim.x >> 6;
    for (int w = 0; w < n_waves; ++w) {
        uint32_t s = lds_wave_sums[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return wave_off + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t lds[SCAN_THREADS / 64];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t total;
    block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of block_sums in place; total -> *total_out (and a copy)
__global__ void __launch_bounds__(1024)
scan_top_kernel(uint32_t* __restrict__ block_sums, uint32_t n_blocks, uint32_t base,
                uint32_t* __restrict__ total_out, uint32_t* __restrict__ total_out2)
{
    __shared__ uint32_t lds[16];
    uint32_t carry = base; // prefix values are absolute indices into the agent arrays
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_blocks ? block_sums[i] : 0;
        uint32_t total;
        uint32_t ex = block_exclusive_scan(v, lds, total);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        if (total_out2) *total_out2 = carry;
    }
}

// out[i] = exclusive prefix; optionally zero the input for the next tick
__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_kernel(uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ block_sums,
                  uint32_t* __restrict__ out, int zero_input)
{
    __shared__ uint32_t lds[SCAN_THREADS / 64];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan(s, lds, total) + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        if (base + k < n) {
            out[base + k] = ex;
            if (zero_input) in[base + k] = 0;
        }
        ex += v[k];
    }
}

// ---- row scan: neighbor_grid_indices in one launch, no inter-workgroup hand-off -----------
// One workgroup per grid row.  Whoever stored a key also counted it on its row (count_key),
// so a row's first index is base + the sum of the row totals before it -- a few hundred
// words every workgroup adds up for itself -- and the cells of the row are then scanned
// locally.  Nothing waits on another workgroup (the decoupled look-back scan this replaces
// spun on its predecessors' status words).  Zeroes the cell counts it consumes; the row
// counts are zeroed by the place kernel that follows (every workgroup here reads them).
// Integrity: a row whose cell counts do not add up to its row count, or a live total above
// the host's bound of the arrays, sets a sticky status bit.
// WAIT (a band's overlapped tick): part of the counts was written by kernels of ANOTHER stream (the unpack of
// the lists on the communication stream) that this launch is not event-ordered behind, so the counts are
// read with agent-scope loads -- coherent whatever this XCD's L2 holds -- instead of relying on nobody
// having pulled those lines in since the launch began.  The plain tick's scan keeps plain loads.
template <bool WAIT> __device__ __forceinline__ uint32_t count_load(const uint32_t* p)
{
    if constexpr (WAIT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

template <bool WAIT>
__global__ void __launch_bounds__(SCAN_THREADS)
scan_rows_kernel(uint32_t* __restrict__ cell_count, const uint32_t* __restrict__ row_count, int32_t row0,
                 int32_t cols, uint32_t base, uint32_t* __restrict__ out, uint32_t* __restrict__ live_out,
                 uint32_t limit, uint32_t* __restrict__ status, const uint32_t* __restrict__ wait_flag,
                 uint32_t wait_seq)
{
    __shared__ uint32_t lds[SCAN_THREADS / 64];
    // A band's overlapped tick: the counts of the lists unpacked on the communication stream are part of
    // this scan's input, and this launch is not ordered behind that stream by an event (a cross-stream
    // event wait between the force launch and this one is ~5 us of idle device; the word has long been
    // written when this launch starts).  Every workgroup looks at the word before its first load.  Seen at
    // the first look -- the rule -- nothing more is needed: the lists were unpacked and written back before
    // this launch began, and its own start has dropped every stale line (an acquire at agent scope here, in
    // 715 workgroups, drops the XCD's whole L2 each time: this kernel 10.2 us instead of 5.6); the counts
    // themselves are read at agent scope (count_load).  Not yet there: wait for it (bounded like
    // edge_wait_kernel), THEN acquire.  The host rides the wait on this launch only while the launch leaves
    // the chip room for the communication stream's kernels (run_row_scan): spinning workgroups that fill
    // every wave slot would keep out the very launches they wait for.
    if (WAIT && wait_flag) {
        if (threadIdx.x == 0 &&
            (int32_t)(__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - wait_seq) < 0) {
            const unsigned long long t0 = wall_clock64();
            while ((int32_t)(__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - wait_seq) < 0) {
                if (wall_clock64() - t0 > 500000000ull) { atomicOr(status, STATUS_EDGE_WAIT); break; }
                __builtin_amdgcn_s_sleep(32);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    const int32_t row = row0 + (int32_t)blockIdx.x;
    uint32_t* in = cell_count + (size_t)row * (size_t)cols;
    uint32_t* o = out + (size_t)row * (size_t)cols;
    // the first chunk of the row's cells is fetched together with the row totals (two
    // independent round trips instead of two dependent ones)
    uint32_t v[4];
    {
        const int32_t first = (int32_t)threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = first + k < cols ? count_load<WAIT>(in + first + k) : 0u;
    }
    // the row totals before this row: four per thread requested at once (1024 rows a round; a loop of single
    // loads made a dependent round trip of every 256 rows)
    uint32_t part = 0;
    for (int32_t r0 = row0 + (int32_t)threadIdx.x; r0 < row; r0 += 4 * SCAN_THREADS) {
        uint32_t t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int32_t r = r0 + k * SCAN_THREADS;
            t[k] = r < row ? count_load<WAIT>(row_count + r) : 0u;
        }
        part += (t[0] + t[1]) + (t[2] + t[3]);
    }
    uint32_t before;
    block_exclusive_scan(part, lds, before);
    uint32_t carry = base + before;
    for (int32_t c0 = 0; c0 < cols; c0 += SCAN_THREADS * 4) {
        const int32_t first = c0 + (int32_t)threadIdx.x * 4;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c0 > 0) v[k] = first + k < cols ? count_load<WAIT>(in + first + k) : 0u;
            s += v[k];
        }
        uint32_t total;
        uint32_t ex = carry + block_exclusive_scan(s, lds, total);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (first + k < cols) {
                o[first + k] = ex;
                in[first + k] = 0;
            }
            ex += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        if (carry - (base + before) != count_load<WAIT>(row_count + row)) atomicOr(status, STATUS_SCAN_MISMATCH);
        if (blockIdx.x == gridDim.x - 1) {
            o[cols] = carry;             // out[(row + 1) * cols]: the end of the last scanned row
            *live_out = carry;
            if (carry > limit) atomicOr(status, STATUS_LIVE_OVERFLOW);
        }
    }
}

// ---- K_PLACE (PEDONI_K_SLOT): rank inside the cell + move ------------------------------------
struct SoA {
    const float2* pos_in; const float4* velx_in; const uint32_t* dest_in;
    float2* pos_out; float4* velx_out; uint32_t* dest_out;
    uint32_t* skey_out;
    int32_t fast;   // PEDONI_MATH_FAST: which form of vl the force kernel expects
};

// moves one agent to its sorted place and fills in its |v| * 0.1 for the pairs in which it
// will be the neighbour (device_math.hpp neighbour_vl)
__device__ __forceinline__ void move_agent(const SoA& a, uint32_t from, uint32_t to, uint32_t packed)
{
    a.pos_out[to] = a.pos_in[from];
    float4 v = a.velx_in[from];
    v.z = a.fast ? neighbour_vl<1>(mk(v.x, v.y)) : neighbour_vl<0>(mk(v.x, v.y));
    a.velx_out[to] = v;
    a.dest_out[to] = a.dest_in[from];
    a.skey_out[to] = packed;
}

// (defined below, beside reorder_kernel)
__device__ __forceinline__ void reorder_body(const uint32_t* __restrict__ key, uint32_t i0, uint32_t n_total,
                                             GridView grid, BandView band, const uint32_t* __restrict__ cs_old,
                                             const uint32_t* __restrict__ cs_new,
                                             const uint32_t* __restrict__ slots,
                                             const SortFlags* __restrict__ flags, uint32_t parity,
                                             uint32_t* __restrict__ cell_count, const SoA& a,
                                             uint32_t first_thread, uint32_t n_threads);

// ---- heaviest tiles first (the force launch's workgroup order) -----------------------------------------
// A force launch is ~2.2 generations of workgroups and ends with a drain: after the last workgroup has been
// dispatched the chip empties over one wave lifetime (tools/force_timeline.py: 28 of 83 us for the uniform
// crowd, 51 of 120 us for the bottleneck's counter-flow, whose densest waves live three times as long as its
// median ones).  What is dispatched LAST should therefore be what ends soonest.  Every wave of a force launch
// stores its lanes' candidate count (ForceArgs.tile_weight: what the pair rounds of the wave follow); the crowd moves centimetres per tick, so that is next tick's cost too.  The next
// sort pass turns the weights into a workgroup order, XCD by XCD (xcd_contiguous_block's chunks: every XCD
// still works through its own contiguous eighth of the agents): a counting sort by weight class, heaviest
// class first (classes a quarter of the mean weight wide, centred so that a uniform crowd is ONE class and
// keeps the plain, L2-friendly ascending order).  Placement only: any weights give a bijection (it is a
// permutation by construction), the results do not depend on it.
// Runs in the first 8 workgroups of the place kernel (one per XCD chunk), which are dispatched first and are
// long done when the launch's other 3900 are: no launch of its own.
__device__ __forceinline__ uint32_t tile_weight_of(const uint32_t* __restrict__ wave_weight, uint32_t tile)
{
    const uint4 w = *reinterpret_cast<const uint4*>(wave_weight + 4u * tile);     // FORCE_WAVES = 4 waves per tile
    return w.x + w.y + w.z + w.w;
}
// A tile is an INDEX range, not a place: its borders drift by tens of agents per tick (agents ahead of it in the
// order change rows), so a jam of ~100 agents near a border is in this tile one tick and in its neighbour the
// next -- 1 % of the bottleneck's tiles change weight by more than half the mean from one tick to the next
// (tools/force_timeline.py).  A tile is therefore ranked by the heaviest of itself and its two neighbours:
// ranking a tile too heavy only starts it early.
// (block_exclusive_scan is defined above; 256 threads)
constexpr uint32_t TILE_LEVELS = 64;      // weight classes, a quarter of the mean weight wide; the one around the mean is
                                          // [0.875, 1.125) x mean: a uniform crowd's tiles all fall into it and keep their order
constexpr uint32_t TILE_CHUNK_MAX = 4096; // tiles per XCD chunk the builder holds in LDS (8.4e6 agents per GPU; beyond: plain order)
__device__ __forceinline__ void build_tile_order(uint32_t xcd, uint32_t n_blocks, const uint32_t* __restrict__ wave_weight,
                                                 uint32_t* __restrict__ order)
{
    __shared__ uint32_t lds[SCAN_THREADS / 64];
    __shared__ uint32_t cursor[TILE_LEVELS];
    __shared__ uint32_t wl[TILE_CHUNK_MAX];
    const uint32_t q = n_blocks / 8u, r = n_blocks % 8u;
    const uint32_t len = q + (xcd < r ? 1u : 0u);
    const uint32_t first = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;     // = xcd_contiguous_block(xcd, n_blocks)
    // the chunk's weights, once, into LDS (one round trip: the loads of a thread are independent), and their sum:
    // tiles are ranked inside their XCD's chunk, so the chunk's own mean is the yardstick
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < len; j += blockDim.x) {
        const uint32_t w = tile_weight_of(wave_weight, first + j);
        wl[j] = w;
        part += w;
    }
    if (threadIdx.x < TILE_LEVELS) cursor[threadIdx.x] = 0;
    uint32_t total;
    block_exclusive_scan(part, lds, total);            // (its barriers publish wl and cursor)
    // class of a tile: heaviest first = class 0; level = round(4 w / mean) (64-bit: no overflow)
    const unsigned long long n64 = len, tot64 = total ? total : 1u;
    auto class_of = [&](uint32_t j) -> uint32_t {
        uint32_t w = wl[j];
        if (j > 0u) w = max(w, wl[j - 1u]);
        if (j + 1u < len) w = max(w, wl[j + 1u]);
        const unsigned long long level = (8ull * w * n64 + tot64) / (2ull * tot64);
        return (TILE_LEVELS - 1u) - (uint32_t)(level < TILE_LEVELS - 1u ? level : TILE_LEVELS - 1u);
    };
    // pass 1: the chunk's tiles per class
    for (uint32_t j = threadIdx.x; j < len; j += blockDim.x) atomicAdd(&cursor[class_of(j)], 1u);
    __syncthreads();
    // exclusive prefix over the classes: where each class starts in the chunk's order
    const uint32_t mine = threadIdx.x < TILE_LEVELS ? cursor[threadIdx.x] : 0u;
    uint32_t all;
    const uint32_t begin = block_exclusive_scan(mine, lds, all);
    if (threadIdx.x < TILE_LEVELS) cursor[threadIdx.x] = begin;
    __syncthreads();
    // pass 2: strips of 256 tiles in ascending order, a tile takes the next place of its class (inside a class
    // the order is ascending from strip to strip and arbitrary inside a strip: placement only)
    for (uint32_t j0 = 0; j0 < len; j0 += blockDim.x) {
        const uint32_t j = j0 + threadIdx.x;
        if (j < len) order[xcd + 8u * atomicAdd(&cursor[class_of(j)], 1u)] = first + j;
        __syncthreads();
    }
}

// Per-cell member counts are already in cell_count when the pass starts: every key that is
// stored -- by the force kernel's tail, by K_KEY or by the halo unpack -- is followed by one
// integer atomicAdd on its cell.  The scan turned them into cell_start (and zeroed them).
//  gather form: agent j scans the old ranges of its NEW cell up to its own index; the members
//    before it = its place in the reference's per-cell list (sfm.rs:66-75).  Ranges of later
//    rows start beyond j and are skipped, so on average half of the 3 x 3 block is read.
//  general form: a provisional slot by a second round of atomics on the (zeroed) counter;
//    K_REORDER puts the cell in order and zeroes the counter again.

struct TileOrder {               // (all null / 0: the force launch keeps the plain XCD-contiguous order)
    const uint32_t* wave_weight; // per wave of the last force launch: the candidates of its 64 lanes
    uint32_t* order;             // out: hardware workgroup -> tile, for a force launch of n_blocks workgroups
    uint32_t n_blocks;
};

template <class DIAG>
__device__ __forceinline__ void place_body(const uint32_t* __restrict__ key, uint32_t i0, uint32_t n_total,
                             GridView grid, BandView band, const uint32_t* __restrict__ cs_old,
                             const uint32_t* __restrict__ cs_new, SortFlags* __restrict__ flags,
                             uint32_t parity, uint32_t* __restrict__ cell_count, SoA a,
                             uint32_t* __restrict__ slots, HaloIn* __restrict__ halo_consumed,
                             uint32_t* __restrict__ row_count, int32_t row0, int32_t row1,
                             uint32_t* __restrict__ status, uint32_t* __restrict__ tickets,
                             uint32_t* __restrict__ done_count, TileOrder tiles, const DIAG diag)
{
    if (diag.off(128)) return;                       // (diagnostics: before anything is read but the arguments)
    // the workgroup order of the force launch that follows this pass, built beside the placement by the
    // launch's first 8 workgroups (block-uniform branch)
    if (tiles.order && blockIdx.x < 8u && blockDim.x == SCAN_THREADS) build_tile_order(blockIdx.x, tiles.n_blocks, tiles.wave_weight, tiles.order);
    // `done_count` != null: the host launches NO reorder kernel after this pass (steady state:
    // nothing appended, not a band) and a general-form pass that only the device knows of -- an agent
    // that moved more than one cell -- is put in order here, by the workgroup that finishes last:
    // every workgroup releases its slot writes (agent scope) and counts itself in; the one that
    // counts last acquires and ranks every cell alone.  Slow and correct, for a case that a finite
    // state cannot reach (|v| dt <= 0.38 m < one cell); the common tick reads one flag and pays nothing.
    const bool collect = done_count != nullptr && flags->far[parity] != 0;
    uint32_t j = i0 + (diag.off(8) ? blockIdx.x : xcd_contiguous_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    // the NEXT tick's flag is raised by this tick's update_states and the next K_KEY; it can
    // be cleared here because every key of this tick has been written and nothing reads it now
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            flags->far[parity ^ 1u] = 0;
            // the agents stored by the device since the last pass (exchanged lists, device
            // spawns) have their keys: mark them consumed
            if (halo_consumed) halo_consumed->n_below = halo_consumed->n_above = halo_consumed->counted = 0;
        }
        // the scan has consumed the row totals: back to zero for the keys of the next pass
        for (int32_t r = row0 + (int32_t)threadIdx.x; r < row1; r += (int32_t)blockDim.x) row_count[r] = 0;
        // (diagnostics build: the tile tickets of the persistent force kernel that follows this pass)
        if (tickets && threadIdx.x < 8u) tickets[threadIdx.x * TICKET_STRIDE] = 0;
    }
    if (diag.off(16)) return;                    // (diagnostics: the launch alone)
    if (diag.off(32)) {                          // (diagnostics: key load + packed-cell store only)
        if (j < n_total) a.skey_out[j] = key[j];
        return;
    }
    if (diag.off(64)) {                          // (diagnostics: the bare record move at j)
        if (j < n_total && key[j] != DEAD) {
            a.pos_out[j] = a.pos_in[j]; a.velx_out[j] = a.velx_in[j]; a.dest_out[j] = a.dest_in[j]; a.skey_out[j] = key[j];
        }
        return;
    }
    const uint32_t c = j < n_total ? key[j] : DEAD;
    if (c != DEAD) {
    uint32_t cy = c / (uint32_t)grid.cols, cx = c - cy * (uint32_t)grid.cols;
    if (!general_cell(flags, parity, band, (int32_t)cy)) {
        // the agent's own record and its cell's start depend on j and c alone: requested BEFORE the
        // rank scan, they arrive while it runs (a latency-bound kernel: one dependent stretch less
        // per wave, 21 -> 19 us).  (Requested even earlier, together with the key: no further gain,
        // 18.9-19.5 us; the first 8 keys of each old range fetched at once instead of one dependent
        // load per candidate: 17.9-18.0 against 18.6-18.8, the tick unchanged -- round 3, not kept.
        // tools/ablate_place.py: switching off the rank scan, the old-range loads or the 56-byte record
        // move changes the launch by < 1 us each; tools/microbench/move_records.hip: the bare move of the
        // same records takes 10 us.)
        const float2 p_in = a.pos_in[j];
        float4 v_in = a.velx_in[j];
        const uint32_t d_in = a.dest_in[j];
        const uint32_t start = cs_new[c];
        CellRanges r{};
        if (!diag.off(4)) r = old_ranges(cs_old, grid, (int32_t)cx, (int32_t)cy);
        uint32_t before = 0;
        if (!diag.off(1)) {
            // four keys per load instruction (the ranges are contiguous; a 16-byte global load needs 4-byte
            // alignment only): the kernel is bound by its number of memory instructions, not by their latency
            // (twelve scalar loads in flight at once, across the three ranges: place 18.3 -> 20.5 us)
            typedef uint32_t key4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t hi = min(r.hi[k], j);
                uint32_t i = r.lo[k];
                for (; i + 4u <= hi; i += 4u) {
                    const key4 q = *reinterpret_cast<const key4*>(key + i);
                    before += (q.x == c ? 1u : 0u) + (q.y == c ? 1u : 0u) + (q.z == c ? 1u : 0u) + (q.w == c ? 1u : 0u);
                }
                for (; i < hi; ++i) before += key[i] == c ? 1u : 0u;
            }
        }
        const uint32_t to = diag.off(1) ? j : start + before;
        // (`to >= n_total` is never taken unless the live count exceeds the host's bound:
        // scan_rows_kernel has raised STATUS_LIVE_OVERFLOW then; do not write past the arrays)
        if (diag.off(2)) {
            if (to < n_total) a.skey_out[to] = pack_cell(cx, cy) + (uint32_t)(p_in.x + v_in.x) + d_in;
        } else if (to < n_total) {                               // = move_agent(a, j, to, ..)
            a.pos_out[to] = p_in;
            v_in.z = a.fast ? neighbour_vl<1>(mk(v_in.x, v_in.y)) : neighbour_vl<0>(mk(v_in.x, v_in.y));
            a.velx_out[to] = v_in;
            a.dest_out[to] = d_in;
            a.skey_out[to] = pack_cell(cx, cy);
        } else atomicOr(status, STATUS_LIVE_OVERFLOW);
    } else {
        const uint32_t to = cs_new[c] + atomicAdd(&cell_count[c], 1u);
        if (to < n_total) slots[to] = j;
        else atomicOr(status, STATUS_LIVE_OVERFLOW);
    }
    }
    if (collect) {                                     // (uniform over the grid: one flag word)
        __shared__ uint32_t is_last;
        __threadfence();                               // release: this workgroup's slots / counters
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t arrived = atomicAdd(done_count, 1u);
            is_last = arrived + 1u == gridDim.x ? 1u : 0u;
            if (is_last) *done_count = 0;              // ready for the next pass
        }
        __syncthreads();
        if (is_last) {
            __threadfence();                           // acquire: every other workgroup's writes
            reorder_body(key, i0, n_total, grid, band, cs_old, cs_new, slots, flags, parity, cell_count, a,
                         threadIdx.x, blockDim.x);
        }
    }
}

__global__ void place_kernel(const uint32_t* __restrict__ key, uint32_t i0, uint32_t n_total,
                             GridView grid, BandView band, const uint32_t* __restrict__ cs_old,
                             const uint32_t* __restrict__ cs_new, SortFlags* __restrict__ flags,
                             uint32_t parity, uint32_t* __restrict__ cell_count, SoA a,
                             uint32_t* __restrict__ slots, HaloIn* __restrict__ halo_consumed,
                             uint32_t* __restrict__ row_count, int32_t row0, int32_t row1,
                             uint32_t* __restrict__ status, uint32_t* __restrict__ tickets,
                             uint32_t* __restrict__ done_count, TileOrder tiles)
{
    place_body(key, i0, n_total, grid, band, cs_old, cs_new, flags, parity, cell_count, a, slots, halo_consumed, row_count, row0, row1, status, tickets, done_count, tiles, NoDiag{});
}

// ---- K_REORDER (general form only) --------------------------------------------------------
// sfm.rs:66-75: an agent's place inside its cell is the number of cell-mates with a smaller
// previous index; the slot list gives those indices in arbitrary (atomic arrival) order.
__device__ __forceinline__ void reorder_body(const uint32_t* __restrict__ key, uint32_t i0, uint32_t n_total,
                                             GridView grid, BandView band, const uint32_t* __restrict__ cs_old,
                                             const uint32_t* __restrict__ cs_new,
                                             const uint32_t* __restrict__ slots,
                                             const SortFlags* __restrict__ flags, uint32_t parity,
                                             uint32_t* __restrict__ cell_count, const SoA& a,
                                             uint32_t first_thread, uint32_t n_threads)
{
    const bool everything = flags->far[parity] != 0;
    if (!everything && !band.sharded) return;
    // sharded, no far mover: only agents landing in the four boundary rows are in general
    // form, and those come from the received lists or from own old rows <= lo+1 / >= hi-2:
    // two index ranges [i0, skip_begin) and [skip_end, n_total)
    uint32_t skip_begin = n_total, skip_end = n_total;
    if (!everything && band.hi - band.lo >= 6) {
        skip_begin = cs_old[(int64_t)(band.lo + 2) * grid.cols];
        skip_end = cs_old[(int64_t)(band.hi - 2) * grid.cols];
        if (skip_end < skip_begin) skip_end = skip_begin;
    }
    const uint32_t skipped = skip_end - skip_begin;
    for (uint32_t t = first_thread; t < n_total - i0 - skipped; t += n_threads) {
        uint32_t i = i0 + t;
        if (i >= skip_begin) i += skipped;
        uint32_t k = key[i];
        if (k == DEAD) continue;
        uint32_t cy = k / (uint32_t)grid.cols, cx = k - cy * (uint32_t)grid.cols;
        if (!general_cell(flags, parity, band, (int32_t)cy)) continue;
        uint32_t base = cs_new[k], end = min(cs_new[k + 1], n_total);
        uint32_t before = 0;
        for (uint32_t j = base; j < end; ++j) before += slots[j] < i ? 1u : 0u;
        if (base + before < n_total) move_agent(a, i, base + before, pack_cell(cx, cy));
        cell_count[k] = 0;   // the provisional-slot counter, back to zero for the next tick's counts
    }
}

__global__ void reorder_kernel(const uint32_t* __restrict__ key, uint32_t i0, uint32_t n_total,
                               GridView grid, BandView band, const uint32_t* __restrict__ cs_old,
                               const uint32_t* __restrict__ cs_new,
                               const uint32_t* __restrict__ slots,
                               const SortFlags* __restrict__ flags, uint32_t parity,
                               uint32_t* __restrict__ cell_count, SoA a)
{
    // grid-stride; launched only when the host knows the pass (or, for a band, its boundary rows) is in
    // general form -- see place_kernel's `inline_reorder` for the case only the device knows of
    reorder_body(key, i0, n_total, grid, band, cs_old, cs_new, slots, flags, parity, cell_count, a,
                 blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// no-grid compaction: survivor i goes to its exclusive flag prefix
__global__ void compact_kernel(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ prefix,
                               uint32_t n_total, const float2* __restrict__ pos_in,
                               const float4* __restrict__ velx_in, const uint32_t* __restrict__ dest_in,
                               float2* __restrict__ pos_out, float4* __restrict__ velx_out,
                               uint32_t* __restrict__ dest_out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total || !flag[i]) return;
    uint32_t p = prefix[i];
    pos_out[p] = pos_in[i];
    velx_out[p] = velx_in[i];      // (the brute-force kernel computes |v| * 0.1 itself)
    dest_out[p] = dest_in[i];
}

// ---- K_FORCE -------------------------------------------------------------------------
struct ForceArgs {
    const float2* pos;   // sorted state (read)
    const float4* velx;  // {vx, vy, |v| * 0.1 (neighbour_vl), desired speed}
    const uint32_t* dest;
    float2* pos_out;     // integrated state (write); may be null for acc-only
    float4* velx_out;
    float2* acc_out;     // optional: accelerations only (no integration)
    const uint32_t* live_count; // absolute end index of the sorted agents
    uint32_t base;              // absolute index of the first sorted agent
    const uint32_t* cell_start;
    const PedoniObstacleDev* obstacles;
    uint32_t n_obstacles;
    FieldView field;
    GridView grid;
    int32_t band_lo, band_hi; // rows whose agents are integrated (others are ghosts)
    // agents handled by this launch: up to two runs of whole grid rows [seg_row[k][0],
    // seg_row[k][1]) (index ranges read off cell_start); seg_row[0][0] < 0 = every sorted agent
    int32_t seg_row[2][2];
    int32_t clear_stale;   // segment launch: surplus threads write DEAD keys to stale slots
    uint32_t* error_word;  // HaloIn.error: bit 2 = a segment was longer than its launch
    int32_t use_grid, use_distance_map;
    // fused K_KEY of the next sort/despawn pass (null = not requested): the agent's next
    // cell key (or DEAD) and the far-mover flag of the next tick's parity
    uint32_t* key_next;
    uint32_t* cell_count; // members per cell of the next pass (one atomicAdd per stored key)
    uint32_t* row_count;  // members per grid row of the next pass (wave-aggregated)
    uint32_t key_end;    // stale slots [live, key_end) get DEAD keys
    SortFlags* flags;
    uint32_t parity_next;
    // per-cell early-out flags (cell_flags_own_kernel above; null = every agent samples): bit dest = the
    // despawn test is certain to pass anywhere in the 3 x 3 cells around, bit 31 = the wall term is +-0
    const uint32_t* cell_flags;
    int32_t xcd_remap; // XCD-contiguous block order (PEDONI_NO_XCD_REMAP=1 turns it off)
    // heaviest tiles first (build_tile_order above): this launch's workgroup -> tile map (null: the plain
    // XCD-contiguous order) and where every wave leaves its weight for the next tick's map (null: nowhere)
    const uint32_t* tile_order;
    uint32_t* tile_weight;
    // edge-first form (force_kernel_queue_edge_first, a band of a sharded run): the tiles holding the agents
    // of the rows below edge_row[0] and from edge_row[1] up are worked on by the FIRST workgroups; when
    // the last of those has stored its results, edge_flag (a device word another stream's edge_wait_kernel
    // polls) is set to edge_seq
    int32_t edge_row[2];
    uint32_t edge_blocks[2];   // hint: workgroups started first for the low / the high edge rows ...
    uint32_t edge_tile_hi;     // ... the latter on tiles [edge_tile_hi, edge_tile_hi + edge_blocks[1])
    uint32_t* edge_counter;
    uint32_t* edge_flag;
    uint32_t edge_seq;
    uint32_t* tickets;  // persistent form: 8 ticket words, TICKET_STRIDE apart (zeroed by the place kernel)
    uint32_t n_tiles;   // persistent form: 64-agent tiles of this launch
    unsigned long long* trace; // diagnostics build only: per-wave records of TraceDiag (kernels_diag.hpp)
    int32_t ablate;            // diagnostics build only: SwitchDiag's bits (kernels_diag.hpp); no product kernel reads it
};

// sfm.rs:117-128: the agent's candidates are three contiguous index ranges, one per grid row iy-1 .. iy+1, columns
// ix-1 .. ix+1 (clamped to the grid), walked in ascending row order.  All six words of neighbor_grid_indices are
// requested at once: a row outside the grid reads row iy's words and is given length 0 (it keeps its place in
// the order and contributes nothing) -- a loop over the rows that exist made dependent round trips of them.
__device__ __forceinline__ void candidate_ranges(const ForceArgs& a, int32_t ix, int32_t iy, uint32_t& r0, uint32_t& n0,
                                                 uint32_t& r1, uint32_t& n1, uint32_t& r2, uint32_t& n2)
{
    const int32_t x_start = max(ix - 1, 0), x_end = min(ix + 1, a.grid.cols - 1);     // :119-120
    uint32_t lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t y = iy - 1 + k;                                                  // :117-118
        const int64_t offset = (int64_t)((y < 0 || y >= a.grid.rows) ? iy : y) * a.grid.cols;
        lo[k] = a.cell_start[offset + x_start];
        hi[k] = a.cell_start[offset + x_end + 1];
    }
    const bool in0 = iy - 1 >= 0, in2 = iy + 1 < a.grid.rows;
    r0 = lo[0]; n0 = in0 ? hi[0] - lo[0] : 0u;
    r1 = lo[1]; n1 = hi[1] - lo[1];
    r2 = lo[2]; n2 = in2 ? hi[2] - lo[2] : 0u;
}

// the early-out flags of cell (ix, iy) -- the cell of a sorted agent, which the sort pass has checked
__device__ __forceinline__ uint32_t cell_flags_of(const ForceArgs& a, int32_t ix, int32_t iy)
{
    if (!a.cell_flags || (uint32_t)ix >= (uint32_t)a.grid.cols || (uint32_t)iy >= (uint32_t)a.grid.rows) return 0u;
    return a.cell_flags[(uint32_t)iy * (uint32_t)a.grid.cols + (uint32_t)ix];
}

// sfm.rs:69 `field.get_potential(destination, pos) > 0.25` at the integrated position (= survives()).
// `far`: the step ended outside the 3 x 3 cells around the agent's old cell, where the flags do not reach.
__device__ __forceinline__ bool despawn_test_passes(const FieldView& f, uint32_t cflags, uint32_t dest, bool far, v2 pos)
{
    const bool certain = !far && dest < CELL_FLAG_MAPS && ((cflags >> dest) & 1u) != 0u && pos.x == pos.x && pos.y == pos.y;
    if (certain) return true;
    if (dest >= f.n_maps) return false;
    const v2 q = field_coord(f, pos);
    return bilinear(potential_map(f, dest), dims_of(f), q.x, q.y) > 0.25f;
}

// goal force, sfm.rs:106-109
template <int MODE>
__device__ __forceinline__ v2 goal_direction(const FieldView& f, v2 pos, uint32_t dest)
{
    v2 q = field_coord(f, pos);
    v2 g = sobel_fast(potential_map(f, dest), dims_of(f), q.x, q.y, nullptr);
    return normalize<0>(g); // exact in both math modes: `e` feeds the field-of-view decision
}

// obstacle force from the distance map, sfm.rs:188-192
template <int MODE>
__device__ __forceinline__ v2 obstacle_force_map(const FieldView& f, v2 pos, const uint64_t* tab)
{
    v2 q = field_coord(f, pos);
    float distance;
    v2 direction = -normalize<MODE>(sobel_fast(f.distance_map, dims_of(f), q.x, q.y, &distance));
    float k = (10.0f * 0.2f) * fexp<MODE>(div_02<MODE>(-distance), tab);
    return direction * k;
}

// obstacle force from explicit wall segments, sfm.rs:193-236
template <int MODE>
__device__ __forceinline__ void obstacle_force_segments(const PedoniObstacleDev* obs, uint32_t n_obs,
                                                        v2 pos, v2& acc, const uint64_t* tab)
{
    // accumulates straight into the agent's running `acc`: fp addition order is part of
    // the result (sfm.rs:226 `acc += force` once per obstacle)
    for (uint32_t o = 0; o < n_obs; ++o) {
        v2 v0 = mk(obs[o].x0, obs[o].y0), v1 = mk(obs[o].x1, obs[o].y1);
        float w = obs[o].width;
        v2 d = v1 - v0;                                      // :197
        float h = length<0>(d);                              // :198
        v2 n = (normalize_or_zero(mk(d.y, -d.x)) * w) * 0.5f; // :199
        v2 df[4];
        df[0] = distance_from_line(pos, v0 + n, v0 - n);     // :200-209
        df[1] = distance_from_line(pos, v1 + n, v1 - n);
        df[2] = distance_from_line(pos, v0 + n, v1 + n);
        df[3] = distance_from_line(pos, v0 - n, v1 - n);
        float ds0 = length<0>(df[0]), ds1 = length<0>(df[1]);
        float ds2 = length<0>(df[2]), ds3 = length<0>(df[3]);
        if (ds0 < w && ds1 < w && ds2 < h && ds3 < h) continue; // :211-217
        v2 dm = df[0]; float min_d = ds0;                    // :218-222 first minimum
        if (ds1 < min_d) { min_d = ds1; dm = df[1]; }
        if (ds2 < min_d) { min_d = ds2; dm = df[2]; }
        if (ds3 < min_d) { min_d = ds3; dm = df[3]; }
        v2 direction = normalize<MODE>(dm);                  // :223
        float k = (10.0f * 0.2f) * fexp<MODE>(div_02<MODE>(-min_d), tab);      // :225
        acc = acc + direction * k;                           // :226
    }
}

// Simple force kernel: one lane per agent, neighbours streamed from L1/L2 in the
// reference's accumulation order (rows ascending, index ascending).  Used for the
// brute-force option path (use_neighbor_grid = false, sfm.rs:157-185) and as an
// independent cross-check of force_kernel_queue in the GPU tests.
template <int MODE>
__global__ void force_kernel_simple(ForceArgs a)
{
    __shared__ uint64_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = EXP2F_TAB[threadIdx.x];
    __syncthreads();

    uint32_t id = a.base + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = *a.live_count;
    if (id >= n) return;

    float2 p = a.pos[id];
    float4 vv = a.velx[id];
    v2 pos = mk(p.x, p.y), vel = mk(vv.x, vv.y);
    float desired_speed = vv.w;
    uint32_t destination = a.dest[id];

    int32_t ix = 0, iy = 0;
    if (a.use_grid) {
        ix = f32_as_i32(pos.x / a.grid.unit);                // sfm.rs:113
        iy = f32_as_i32(pos.y / a.grid.unit);
        if (iy < a.band_lo || iy >= a.band_hi) {             // ghost row: never integrated
            if (a.pos_out) {                                 // NaN position = "not mine"; the
                float qn = __builtin_nanf("");               // next sort/despawn pass drops it
                a.pos_out[id] = make_float2(qn, qn);
                a.velx_out[id] = vv;
            }
            return;
        }
    }

    v2 acc = mk(0.0f, 0.0f);                                 // :104
    v2 e = goal_direction<MODE>(a.field, pos, destination);  // :107-108
    acc = acc + vdiv<MODE>(e * desired_speed - vel, 0.5f);   // :109

    if (a.use_grid) {                                        // :112-156
        int32_t y_start = max(iy - 1, 0), y_end = min(iy + 1, a.grid.rows - 1);
        int32_t x_start = max(ix - 1, 0), x_end = min(ix + 1, a.grid.cols - 1);
        for (int32_t y = y_start; y <= y_end; ++y) {
            int64_t offset = (int64_t)y * a.grid.cols;
            uint32_t i_start = a.cell_start[offset + x_start];
            uint32_t i_end = a.cell_start[offset + x_end + 1];
            for (uint32_t i = i_start; i < i_end; ++i) {
                if (i != id) {
                    float2 pi = a.pos[i];
                    float4 vi = a.velx[i];
                    pair_force<MODE>(pos, e, mk(pi.x, pi.y), mk(vi.x, vi.y), acc, tab);
                }
            }
        }
    } else {                                                 // :157-185
        for (uint32_t i = a.base; i < n; ++i) {
            if (i != id) {
                float2 pi = a.pos[i];
                float4 vi = a.velx[i];
                pair_force<MODE>(pos, e, mk(pi.x, pi.y), mk(vi.x, vi.y), acc, tab);
            }
        }
    }

    if (a.use_distance_map) acc = acc + obstacle_force_map<MODE>(a.field, pos, tab);
    else obstacle_force_segments<MODE>(a.obstacles, a.n_obstacles, pos, acc, tab);

    if (a.acc_out) { a.acc_out[id] = make_float2(acc.x, acc.y); return; }

    // integrator, sfm.rs:245-254
    v2 vel_prev = vel;
    vel = vel + acc * 0.1f;
    float max_len = desired_speed * 1.3f;
    float length_sq = dot(vel, vel);
    if (length_sq > max_len * max_len) {                     // glam clamp_length_max
        v2 q = vdiv<MODE>(vel, fsqrt<MODE>(length_sq));
        vel = mk(max_len * q.x, max_len * q.y);
    }
    pos = pos + (vel + vel_prev) * 0.05f;
    a.pos_out[id] = make_float2(pos.x, pos.y);
    a.velx_out[id] = make_float4(vel.x, vel.y, 0.0f, desired_speed);   // (vl: filled by the next sort pass)
}

// ---- K_FORCE, wave-queue form (grid path) ----------------------------------------------
// One lane owns one agent (sorted order, so a wave covers ~32 neighbouring cells), but
// the expensive pair evaluation is decoupled from ownership:
//   phase 1  every lane walks its own candidate list (3 contiguous index ranges, rows
//            y-1..y+1) SLOTS candidates at a time and does only the cutoff test
//            (|d|^2 > 4 -> skip, sfm.rs:133), on positions only.  Survivors are compacted
//            with ballot + mbcnt into a per-wave queue in LDS: {dx, dy} and one word holding
//            the neighbour's index (26 bits) and the owner lane (6 bits): 12 bytes an entry.
//   phase 2  the queue is drained 64 entries at a time: every lane fetches its neighbour's
//            {vx, vy, |v| * 0.1} (one 16-byte load, issued first; the distance / direction
//            arithmetic that does not need it runs underneath), evaluates one pair force (4
//            sqrt, 6 div, 1 exp) with no divergence and overwrites {dx, dy} with the result.
//   phase 3  each owner adds its results in candidate order -- rows ascending, index
//            ascending, exactly the reference's `acc += force` sequence -- so the sum is
//            bit-identical to the serial loop although pairs were evaluated in parallel.
// The queue is private to a wave (LDS operations of one wave retire in order), so the
// loop has no workgroup barrier.
// Tuning notes (N = 1e6, exact mode): 6 slots per batch = 28 KB LDS per block = 5 waves/SIMD
// at 82 VGPR: 0.141 ms; 8 slots (4 waves) 0.151; 12-16 slots (3-2 waves) 0.22; forcing 6
// waves (<= 80 VGPR, 12 B spill) 0.141: the kernel sits on a ~80 % VALU-busy plateau.
// Carrying the < 64 left-over queue entries of a batch into the next one (full phase-2
// rounds only, -20 % rounds) was built, verified bit-exact and measured: 0.142 ms, no gain
// -- a partly filled round is cheap (idle half-waves are skipped) and the carry bookkeeping
// costs what it saves.  Ablation of the final kernel: pairs ~95 us, field stencils ~22 us,
// loads / fused key / integrator / stores ~28 us.
#ifndef PEDONI_FORCE_THREADS
#define PEDONI_FORCE_THREADS 256     // (tools/ab_repeat.sh "-DPEDONI_FORCE_THREADS=128": see profiles/r03_force_threads_ab.txt)
#endif
constexpr int FORCE_THREADS = PEDONI_FORCE_THREADS;
constexpr int FORCE_WAVES = FORCE_THREADS / 64;

// One tile = the 64 agents of one wave: sorted indices base + 64 * tile + lane.  `queue` / `who` are
// the calling wave's own LDS queue, `tab` the block's copy of the exp table.
// EXIT: called by every lane exactly once, where it leaves the tile (the lanes of a wave leave at up to four
// places; nothing runs after the tile function that would need them back together -- the mask to restore
// would be two more SGPRs held across the whole tile, which at the 7-wave budget spills).  The default does nothing.
struct NoExit { __device__ __forceinline__ void operator()() const {} };

template <int MODE, int SLOTS, class DIAG = NoDiag, class EXIT = NoExit>
__device__ __forceinline__ void force_queue_tile(const ForceArgs& a, const uint32_t t0, float2* __restrict__ queue,
                                                 uint32_t* __restrict__ who, const uint64_t* __restrict__ tab,
                                                 const EXIT on_exit = EXIT{})
{
    DIAG diag(a, t0 >> 6);           // (NoDiag in every product kernel: see the top of this file)

    const uint32_t lane = threadIdx.x & 63u;
    uint32_t id = a.base + t0;
    uint32_t n = *a.live_count;
    if (a.seg_row[0][0] >= 0) {                    // row-segment launch (sharded overlap)
        uint32_t t = t0;
        uint32_t b0 = a.cell_start[(int64_t)a.seg_row[0][0] * a.grid.cols];
        uint32_t e0 = a.cell_start[(int64_t)a.seg_row[0][1] * a.grid.cols];
        uint32_t b1 = a.cell_start[(int64_t)a.seg_row[1][0] * a.grid.cols];
        uint32_t e1 = a.cell_start[(int64_t)a.seg_row[1][1] * a.grid.cols];
        if (t < e0 - b0) { id = b0 + t; n = e0; }
        else { id = b1 + (t - (e0 - b0)); n = e1; }
        uint32_t need = (e0 - b0) + (e1 - b1);
        if (t == 0 && need > gridDim.x * blockDim.x) atomicOr(a.error_word, 4u);
        if (t >= need) {
            // surplus thread: clear the key of a slot left behind by a despawned agent
            uint32_t stale = *a.live_count + (t - need);
            if (a.clear_stale && a.key_next && stale < a.key_end) a.key_next[stale] = DEAD;
            id = 0xffffffffu; n = 0;
        }
    }
    bool valid = id < n;

    v2 pos = mk(0.0f, 0.0f), vel = mk(0.0f, 0.0f), acc = mk(0.0f, 0.0f), e = mk(0.0f, 0.0f);
    float4 vv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float desired_speed = 0.0f;
    uint32_t r0 = 0, r1 = 0, r2 = 0, n0 = 0, n1 = 0, n2 = 0;
    bool ghost = false;
    int32_t ix = 0, iy = 0;   // the agent's cell (kept for the far-mover test of the tail)
    if (valid) {
        const float2 p = a.pos[id];
        vv = a.velx[id];
        pos = mk(p.x, p.y);
        vel = mk(vv.x, vv.y);
        desired_speed = vv.w;
        const uint32_t destination = a.dest[id];
        ix = f32_as_i32(pos.x / a.grid.unit);                    // sfm.rs:113
        iy = f32_as_i32(pos.y / a.grid.unit);
        ghost = iy < a.band_lo || iy >= a.band_hi;
        if (!ghost) {
            candidate_ranges(a, ix, iy, r0, n0, r1, n1, r2, n2);      // :117-120 (requested ahead of the goal stencil's texels)
            if (diag.off(1)) e = mk(1.0f, 0.0f);
            else e = goal_direction<MODE>(a.field, pos, destination); // :107-108
            acc = acc + vdiv<MODE>(e * desired_speed - vel, 0.5f); // :109
        }
    }
    const uint32_t cnt = diag.off(4) ? 0u : n0 + n1 + n2;
    uint32_t max_cnt = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_cnt = max(max_cnt, (uint32_t)__shfl_xor((int)max_cnt, off, 64));
    max_cnt = __builtin_amdgcn_readfirstlane(max_cnt);
    // this wave's weight for the next tick's workgroup order: its candidates, all lanes together (what the pair
    // rounds of phase 2 follow; steadier from tick to tick than the fullest lane's count, which jumps by 60 %
    // when a clump of agents crosses a tile boundary)
    if (a.tile_weight) {
        uint32_t sum_cnt = cnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum_cnt += (uint32_t)__shfl_xor((int)sum_cnt, off, 64);
        if (lane == 0u) a.tile_weight[t0 >> 6] = sum_cnt;
    }

    // candidate s of this lane: index r0 + s, r1 + (s - n0) or r2 + (s - n0 - n1)
    const uint32_t n01 = n0 + n1, r1s = r1 - n0, r2s = r2 - n01;
    // a lane without a candidate in some slot reads its own record instead (always a valid
    // address, already in cache): phase 1 then has no divergent branch around its loads
    const uint32_t id_safe = valid ? id : a.base;

    diag.lap(0);
    for (uint32_t base = 0; base < max_cnt; base += SLOTS) {
        // ---- phase 1: cutoff test + compaction ---------------------------------------
        // three unrolled sub-passes so that the SLOTS position loads, then the SLOTS
        // velocity loads, are all in flight together
        uint32_t qlen = 0;       // wave-uniform
        uint32_t at_of[SLOTS];   // queue entry of candidate k: its slot, or this lane's dump entry
        uint32_t idx[SLOTS];
        float2 d[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            uint32_t s = base + k;
            uint32_t i = s + (s < n0 ? r0 : (s < n01 ? r1s : r2s));
            idx[k] = s < cnt ? i : id_safe;
            // (indices are < 2^26: a 32-bit byte offset beside the uniform base address)
            d[k] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.pos) + (idx[k] << 3));
        }
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            float dx = pos.x - d[k].x;                            // :131
            float dy = pos.y - d[k].y;
            float d2 = (dx * dx) + (dy * dy);                     // :132
            // :130,133 (idx == id also marks "no candidate in this slot").  The two lane masks are
            // taken straight from the compares (hipcc re-materialises a ballot of a bool with two
            // uses through a VGPR: two more VALU instructions per slot) and so is the select.
            unsigned long long m_near, m_other;
            asm("v_cmp_nlt_f32_e64 %0, 4.0, %1" : "=s"(m_near) : "v"(d2));            // !(d2 > 4): NaN passes, as upstream
            asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(m_other) : "v"(idx[k]), "v"(id_safe));
            const unsigned long long mask = m_near & m_other;
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            uint32_t at;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(at) : "v"((uint32_t)(SLOTS * 64) + lane), "v"(qlen + before), "s"(mask));
            at_of[k] = at;
            queue[at] = make_float2(dx, dy);
            who[at] = idx[k] | (lane << 26);
            qlen += (uint32_t)__popcll(mask);
        }
        // the dump entry now reads -0: phase 3 adds it for every slot that did not pass, and
        // x + (-0) == x bit for bit for every x (+-0, denormals -- preserved in this build -- and NaN
        // included), so the ordered sum needs neither a test nor a select per slot
        queue[(uint32_t)(SLOTS * 64) + lane] = make_float2(-0.0f, -0.0f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        diag.lap(1);

        // ---- phase 2: one pair force per lane, no divergence ------------------------------
        for (uint32_t q0 = 0; q0 < qlen; q0 += 64) {
            uint32_t q = q0 + lane;
            const bool busy = q < qlen;
            const uint32_t w = busy ? who[q] : (id_safe | (lane << 26));
            // the neighbour's velocity and |v| * 0.1 (sfm.rs:140,144): one 16-byte load
            // (PEDONI_ABLATE & 8, diagnostics: every lane reads the velocity record of ITS OWN agent --
            // a coalesced, cached load in place of the gather; results wrong, arithmetic the same)
            const float4 vn = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.velx) +
                                                               ((diag.off(8) ? id_safe : (w & 0x03ffffffu)) << 4));
            // the owner's goal direction, straight from its registers (every lane takes part)
            const int own = (int)(w >> 26);
            const float eo_x = __shfl(e.x, own, 64), eo_y = __shfl(e.y, own, 64);
            if (busy) {
                float2 en = queue[q];
                // the force itself (the owner's `acc += force` of sfm.rs:153 happens in phase 3)
                // (PEDONI_ABLATE & 16, diagnostics: no pair arithmetic -- the loads, the queue and the
                // ordered sums stay)
                v2 f = mk(en.x + vn.x + eo_x, en.y + vn.y + eo_y + vn.z);
                if (!diag.off(16)) f = pair_force_value<MODE>(mk(en.x, en.y), mk(eo_x, eo_y), mk(vn.x, vn.y), vn.z, tab);
                queue[q] = make_float2(f.x, f.y);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        diag.lap(2);

        // ---- phase 3: ordered accumulation (sfm.rs:153) ---------------------------------
        // all SLOTS entries are fetched first (a slot that did not pass reads the lane's dump
        // entry, -0), then added in candidate order: no branch, no select, no LDS round trip per slot
        float2 fr[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) fr[k] = queue[at_of[k]];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) acc = acc + mk(fr[k].x, fr[k].y);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        diag.lap(3);
    }

    if (!valid) {
        // slot of a despawned agent (whole-array launches only: segments end at live agents)
        if (a.key_next && a.seg_row[0][0] < 0 && id < a.key_end) a.key_next[id] = DEAD;
    }
    // (the epilogue below runs per lane; the trace is flushed by lane 0 wherever it leaves)
    if (!valid) {
        diag.flush(); on_exit();
        return;
    }
    if (ghost) {                                                  // ghost row: never integrated
        if (a.pos_out) {                                          // NaN position = "not mine"; the
            float qn = __builtin_nanf("");                        // next sort/despawn pass drops it
            a.pos_out[id] = make_float2(qn, qn);
            a.velx_out[id] = vv;
            if (a.key_next) a.key_next[id] = DEAD;
        }
        diag.flush(); on_exit();
        return;
    }

    // (the despawn test's destination and map pointer, requested here so that they travel beside the
    // wall texels instead of forming two dependent stretches of their own after the integrator)
    uint32_t dest_k = 0;
    if (a.key_next) dest_k = a.dest[id];
    // The agent's cell once more, from its position (the same two divisions as in the prologue): carried
    // round the pair loop the two coordinates are two more live VGPRs in a kernel that has none to spare
    // (16 bytes of scratch in the 7-wave build); the asm keeps the compiler from re-using the first result.
    {
        float px = pos.x, py = pos.y;
        asm volatile("" : "+v"(px), "+v"(py));
        ix = f32_as_i32(px / a.grid.unit);
        iy = f32_as_i32(py / a.grid.unit);
    }
    // the early-out flags of that cell (one word, neighbouring lanes read neighbouring words)
    const uint32_t cflags = cell_flags_of(a, ix, iy);
    if (diag.off(2)) {}
    else if (a.use_distance_map) {
        // sfm.rs:188-192.  Flagged cell: exp(-distance / 0.2) is exactly 0 and the direction finite, the term
        // is (+-0, +-0), and acc + (+-0) == acc bit for bit -- unless a component of acc is itself +-0 (or
        // NaN: kept on the sampled path too), where the sign of the zero added would show.
        const bool wall_is_zero = (cflags & CELL_FLAG_WALL) != 0u && __builtin_fabsf(acc.x) > 0.0f && __builtin_fabsf(acc.y) > 0.0f;
        if (!wall_is_zero) acc = acc + obstacle_force_map<MODE>(a.field, pos, tab);
    }
    else {
        // sfm.rs:193-236: `acc += force` once per obstacle; flagged cell: every one of those is (+-0, +-0) or skipped
        const bool walls_are_zero = (cflags & CELL_FLAG_WALL) != 0u && __builtin_fabsf(acc.x) > 0.0f && __builtin_fabsf(acc.y) > 0.0f;
        if (!walls_are_zero) obstacle_force_segments<MODE>(a.obstacles, a.n_obstacles, pos, acc, tab);
    }

    if (a.acc_out) { a.acc_out[id] = make_float2(acc.x, acc.y); diag.flush(); on_exit(); return; }

    // integrator, sfm.rs:245-254
    v2 vel_prev = vel;
    vel = vel + acc * 0.1f;
    float max_len = desired_speed * 1.3f;
    float length_sq = dot(vel, vel);
    if (length_sq > max_len * max_len) {                          // glam clamp_length_max
        v2 q = vdiv<MODE>(vel, fsqrt<MODE>(length_sq));
        vel = mk(max_len * q.x, max_len * q.y);
    }
    pos = pos + (vel + vel_prev) * 0.05f;
    a.pos_out[id] = make_float2(pos.x, pos.y);
    a.velx_out[id] = make_float4(vel.x, vel.y, 0.0f, desired_speed);   // (vl: filled by the next sort pass)

    // fused K_KEY for the next tick: same arithmetic as key_kernel on the new position (the
    // potential texels are the ones the goal stencil just touched, so they come from L1/L2)
    if (a.key_next) {
        uint32_t k = DEAD;
        int32_t cx = 0, cy = 0;
        if (cell_xy(a.grid, pos, cx, cy) && cy >= a.band_lo - 1 && cy <= a.band_hi) {
            const bool far = abs(cx - ix) > 1 || abs(cy - iy) > 1;
            // = survives(a.field, pos, dest_k): certain without a sample when the flag of the agent's map is
            // set and the step ended inside the 3 x 3 cells the flag speaks for, at a position that is a number
            if (diag.off(32) || despawn_test_passes(a.field, cflags, dest_k, far, pos)) {
                k = (uint32_t)cy * (uint32_t)a.grid.cols + (uint32_t)cx;
                if (far) atomicOr(&a.flags->far[a.parity_next], 1u);
            }
        }
        a.key_next[id] = k;
        if (diag.off(64)) { if (k != DEAD) atomicAdd(&a.cell_count[k], 1u); }     // (diagnostics: no row counts)
        else if (diag.off(128)) { if (k == 0xfffffffeu) atomicAdd(&a.cell_count[k], 1u); }   // (no counts at all)
        else count_key(a.cell_count, a.row_count, k != DEAD, k, (uint32_t)cy);
    }
    diag.flush(); on_exit();
}

// LDS of one force-kernel block: the exp table and one pair queue per wave
// (+ 64 entries per wave: lane l of a slot that did not pass writes entry SLOTS * 64 + l, so
// the queue writes of phase 1 need no branch)
#define PEDONI_FORCE_LDS(SLOTS)                                                                        \
    __shared__ uint64_t tab[32];                                                                       \
    __shared__ float2 queue_all[FORCE_WAVES][SLOTS * 64 + 64]; /* {dx, dy} in, {fx, fy} out */         \
    __shared__ uint32_t who_all[FORCE_WAVES][SLOTS * 64 + 64]; /* neighbour index | owner lane << 26 */ \
    if (threadIdx.x < 32) tab[threadIdx.x] = EXP2F_TAB[threadIdx.x];                                   \
    __syncthreads()

// one tile per wave, tiles dealt by the hardware's workgroup order (XCD-contiguous by default)
template <int MODE, int SLOTS, class DIAG = NoDiag>
__device__ __forceinline__ void force_queue_body(const ForceArgs& a)
{
    PEDONI_FORCE_LDS(SLOTS);
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t block = a.tile_order ? a.tile_order[blockIdx.x] : (a.xcd_remap ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x);
    force_queue_tile<MODE, SLOTS, DIAG>(a, block * blockDim.x + threadIdx.x, queue_all[wave], who_all[wave], tab);
}

// ---- K_FORCE for SMALL crowds: G lanes per agent ----------------------------
// With ~1e5 agents (BASELINE C2) the one-lane-per-agent kernel puts one or two waves on a SIMD and
// its launch lasts as long as its HEAVIEST wave: a wave is as slow as its fullest lane (70 candidate
// slots in C2's densest cells), and nothing else is there to run meanwhile.  Here a wave owns 64 / G
// agents and the G lanes of an agent share its work:
//   phase 1  lane g of the group tests the candidate slots s = g, g + G, g + 2 G ... (the fullest
//            lane's walk is G times shorter); survivors go to the same per-wave queue.
//   phase 2  unchanged: the queue is drained 64 pairs at a time whoever queued them.
//   phase 3  every lane of the group forms the SAME ordered sum: for each batch position k the
//            group's G results are taken in slot order g = 0 .. G-1 from the neighbouring lanes by
//            DPP (quad_perm, no LDS traffic) -- acc += f(slot G k), acc += f(slot G k + 1), ... --
//            exactly the reference's `acc += force` sequence (sfm.rs:153), so the bits are the
//            one-lane kernel's.  A slot without a pair contributes -0 (x + (-0) == x for every x).
// Everything else -- the agent's loads, both stencils, the integrator, the next key -- is computed
// by all G lanes alike (same inputs, same results); lane 0 of the group stores and counts.  G times
// the waves, each with a G times shorter critical path: the crowd's makespan, not its instruction
// count, is what a small launch pays for.
template <int G> __device__ __forceinline__ float group_lane(float v, int g)
{
    // value of lane g of this lane's group (G = 2: pairs inside a quad; G = 4: the quad)
    static_assert(G == 2 || G == 4, "groups of 2 or 4 lanes");
    int r;
    const int x = __float_as_int(v);
    if constexpr (G == 4) {
        switch (g) {
        case 0: r = __builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true); break;   // quad_perm:[0,0,0,0]
        case 1: r = __builtin_amdgcn_mov_dpp(x, 0x55, 0xf, 0xf, true); break;   // [1,1,1,1]
        case 2: r = __builtin_amdgcn_mov_dpp(x, 0xaa, 0xf, 0xf, true); break;   // [2,2,2,2]
        default: r = __builtin_amdgcn_mov_dpp(x, 0xff, 0xf, 0xf, true); break;  // [3,3,3,3]
        }
    } else {
        if (g == 0) r = __builtin_amdgcn_mov_dpp(x, 0xa0, 0xf, 0xf, true);      // [0,0,2,2]
        else r = __builtin_amdgcn_mov_dpp(x, 0xf5, 0xf, 0xf, true);             // [1,1,3,3]
    }
    return __int_as_float(r);
}

template <int MODE, int SLOTS, int G, class DIAG = NoDiag>
__device__ __forceinline__ void force_queue_tile_group(const ForceArgs& a, const uint32_t tile, float2* __restrict__ queue,
                                                       uint32_t* __restrict__ who, const uint64_t* __restrict__ tab)
{
    DIAG diag(a, tile);
    constexpr uint32_t PER_WAVE = 64u / (uint32_t)G;
    const uint32_t lane = threadIdx.x & 63u, sub = lane & (uint32_t)(G - 1);
    const bool writer = sub == 0;                 // the group's lane that stores and counts
    uint32_t id = a.base + tile * PER_WAVE + lane / (uint32_t)G;
    uint32_t n = *a.live_count;
    if (a.seg_row[0][0] >= 0) {                   // row-segment launch (the edge rows of a sharded tick)
        const uint32_t t = tile * PER_WAVE + lane / (uint32_t)G;      // agent slot of this launch
        const uint32_t b0 = a.cell_start[(int64_t)a.seg_row[0][0] * a.grid.cols];
        const uint32_t e0 = a.cell_start[(int64_t)a.seg_row[0][1] * a.grid.cols];
        const uint32_t b1 = a.cell_start[(int64_t)a.seg_row[1][0] * a.grid.cols];
        const uint32_t e1 = a.cell_start[(int64_t)a.seg_row[1][1] * a.grid.cols];
        if (t < e0 - b0) { id = b0 + t; n = e0; }
        else { id = b1 + (t - (e0 - b0)); n = e1; }
        const uint32_t need = (e0 - b0) + (e1 - b1);
        if (t == 0 && writer && need > gridDim.x * (blockDim.x / (uint32_t)G)) atomicOr(a.error_word, 4u);
        if (t >= need) {
            const uint32_t stale = *a.live_count + (t - need);
            if (writer && a.clear_stale && a.key_next && stale < a.key_end) a.key_next[stale] = DEAD;
            id = 0xffffffffu; n = 0;
        }
    }
    const bool valid = id < n;

    v2 pos = mk(0.0f, 0.0f), vel = mk(0.0f, 0.0f), acc = mk(0.0f, 0.0f), e = mk(0.0f, 0.0f), wall = mk(0.0f, 0.0f);
    float4 vv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float desired_speed = 0.0f;
    uint32_t r0 = 0, r1 = 0, r2 = 0, n0 = 0, n1 = 0, n2 = 0;
    bool ghost = false;
    int32_t ix = 0, iy = 0;
    if (valid) {
        float2 p = a.pos[id];
        vv = a.velx[id];
        pos = mk(p.x, p.y);
        vel = mk(vv.x, vv.y);
        desired_speed = vv.w;
        uint32_t destination = a.dest[id];
        ix = f32_as_i32(pos.x / a.grid.unit);                    // sfm.rs:113
        iy = f32_as_i32(pos.y / a.grid.unit);
        ghost = iy < a.band_lo || iy >= a.band_hi;
        if (!ghost) {
            // ONE stencil evaluation per lane: lane 1 of the group samples the distance map (the wall
            // force, sfm.rs:188-192), the others the agent's potential map (the goal direction,
            // :107-108) -- the same 4 x 4 patch code on a different map pointer, so the two stencils of
            // an agent run side by side instead of one after the other.  Both results are then
            // handed round the group by DPP.  (The explicit-segment wall path stays in the epilogue.)
            candidate_ranges(a, ix, iy, r0, n0, r1, n1, r2, n2);      // :117-120 (requested ahead of the stencil's texels)
            const bool wall_lane = sub == 1 && a.use_distance_map;
            const float* map = wall_lane ? a.field.distance_map
                                         : (destination < a.field.n_maps ? potential_map(a.field, destination) : a.field.distance_map);
            const v2 q = field_coord(a.field, pos);
            float centre;
            const v2 g = sobel_fast(map, dims_of(a.field), q.x, q.y, &centre);
            const v2 e_mine = normalize<0>(g);                                        // goal lanes (exact in both modes)
            const v2 w_dir = -normalize<MODE>(g);                                     // wall lane: obstacle_force_map
            const float w_k = (10.0f * 0.2f) * fexp<MODE>(div_02<MODE>(-centre), tab);
            e = mk(group_lane<G>(e_mine.x, 0), group_lane<G>(e_mine.y, 0));
            wall = mk(group_lane<G>(w_dir.x * w_k, 1), group_lane<G>(w_dir.y * w_k, 1));
            acc = acc + vdiv<MODE>(e * desired_speed - vel, 0.5f); // :109
        }
    }
    const uint32_t cnt = n0 + n1 + n2;            // the agent's candidates: the same on all G lanes
    uint32_t max_cnt = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_cnt = max(max_cnt, (uint32_t)__shfl_xor((int)max_cnt, off, 64));
    max_cnt = __builtin_amdgcn_readfirstlane(max_cnt);
    const uint32_t n01 = n0 + n1, r1s = r1 - n0, r2s = r2 - n01;
    const uint32_t id_safe = valid ? id : a.base;

    diag.lap(0);
    // one batch = SLOTS slots per lane = G * SLOTS consecutive slots of the agent
    for (uint32_t base = 0; base < max_cnt; base += (uint32_t)(G * SLOTS)) {
        // ---- phase 1: this lane's share of the batch: slots base + sub, base + sub + G, ... ----
        uint32_t qlen = 0;
        uint32_t at_of[SLOTS];
        uint32_t idx[SLOTS];
        float2 d[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            uint32_t s = base + (uint32_t)(k * G) + sub;
            uint32_t i = s + (s < n0 ? r0 : (s < n01 ? r1s : r2s));
            idx[k] = s < cnt ? i : id_safe;
            d[k] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.pos) + (idx[k] << 3));
        }
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            float dx = pos.x - d[k].x;                            // :131
            float dy = pos.y - d[k].y;
            float d2 = (dx * dx) + (dy * dy);                     // :132
            unsigned long long m_near, m_other;
            asm("v_cmp_nlt_f32_e64 %0, 4.0, %1" : "=s"(m_near) : "v"(d2));            // !(d2 > 4): NaN passes, as upstream
            asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(m_other) : "v"(idx[k]), "v"(id_safe));
            const unsigned long long mask = m_near & m_other;
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            uint32_t at;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(at) : "v"((uint32_t)(SLOTS * 64) + lane), "v"(qlen + before), "s"(mask));
            at_of[k] = at;
            queue[at] = make_float2(dx, dy);
            who[at] = idx[k] | (lane << 26);
            qlen += (uint32_t)__popcll(mask);
        }
        queue[(uint32_t)(SLOTS * 64) + lane] = make_float2(-0.0f, -0.0f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        diag.lap(1);

        // ---- phase 2: one pair force per lane (as in the one-lane kernel) ---------------------
        for (uint32_t q0 = 0; q0 < qlen; q0 += 64) {
            uint32_t q = q0 + lane;
            const bool busy = q < qlen;
            const uint32_t w = busy ? who[q] : (id_safe | (lane << 26));
            const float4 vn = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.velx) + ((w & 0x03ffffffu) << 4));
            const int own = (int)(w >> 26);
            const float eo_x = __shfl(e.x, own, 64), eo_y = __shfl(e.y, own, 64);
            if (busy) {
                float2 en = queue[q];
                v2 f = pair_force_value<MODE>(mk(en.x, en.y), mk(eo_x, eo_y), mk(vn.x, vn.y), vn.z, tab);
                queue[q] = make_float2(f.x, f.y);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        diag.lap(2);

        // ---- phase 3: the group's ordered sum, formed alike on each of its lanes (sfm.rs:153) ----
        float2 fr[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) fr[k] = queue[at_of[k]];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
#pragma unroll
            for (int g = 0; g < G; ++g)           // slot base + k G + g was lane g's k-th
                acc = acc + mk(group_lane<G>(fr[k].x, g), group_lane<G>(fr[k].y, g));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        diag.lap(3);
    }

    if (!valid) {
        // slot of a despawned agent (whole-array launches only: segments end at live agents)
        if (writer && a.key_next && a.seg_row[0][0] < 0 && id < a.key_end) a.key_next[id] = DEAD;
        diag.flush();
        return;
    }
    if (ghost) {                                                  // ghost row: never integrated
        if (writer && a.pos_out) {
            float qn = __builtin_nanf("");
            a.pos_out[id] = make_float2(qn, qn);
            a.velx_out[id] = vv;
            if (a.key_next) a.key_next[id] = DEAD;
        }
        diag.flush();
        return;
    }
    uint32_t dest_k = 0;
    if (a.key_next) dest_k = a.dest[id];
    const uint32_t cflags = cell_flags_of(a, ix, iy);             // (the despawn bits; the wall stencil ran beside the goal's)
    if (a.use_distance_map) acc = acc + wall;                     // (= obstacle_force_map: direction * k, lane 1's)
    else if (!((cflags & CELL_FLAG_WALL) != 0u && __builtin_fabsf(acc.x) > 0.0f && __builtin_fabsf(acc.y) > 0.0f))
        obstacle_force_segments<MODE>(a.obstacles, a.n_obstacles, pos, acc, tab);   // (flagged cell: every term +-0, see force_queue_tile)

    if (a.acc_out) { if (writer) a.acc_out[id] = make_float2(acc.x, acc.y); diag.flush(); return; }

    // integrator, sfm.rs:245-254
    v2 vel_prev = vel;
    vel = vel + acc * 0.1f;
    float max_len = desired_speed * 1.3f;
    float length_sq = dot(vel, vel);
    if (length_sq > max_len * max_len) {                          // glam clamp_length_max
        v2 q = vdiv<MODE>(vel, fsqrt<MODE>(length_sq));
        vel = mk(max_len * q.x, max_len * q.y);
    }
    pos = pos + (vel + vel_prev) * 0.05f;
    if (writer) {
        a.pos_out[id] = make_float2(pos.x, pos.y);
        a.velx_out[id] = make_float4(vel.x, vel.y, 0.0f, desired_speed);
    }
    if (a.key_next) {
        uint32_t k = DEAD;
        int32_t cx = 0, cy = 0;
        if (cell_xy(a.grid, pos, cx, cy) && cy >= a.band_lo - 1 && cy <= a.band_hi) {
            const bool far = abs(cx - ix) > 1 || abs(cy - iy) > 1;
            if (despawn_test_passes(a.field, cflags, dest_k, far, pos)) {
                k = (uint32_t)cy * (uint32_t)a.grid.cols + (uint32_t)cx;
                if (writer && far) atomicOr(&a.flags->far[a.parity_next], 1u);
            }
        }
        if (writer) a.key_next[id] = k;
        count_key(a.cell_count, a.row_count, writer && k != DEAD, k, (uint32_t)cy);
    }
    diag.flush();
}

template <int MODE, int SLOTS, int G>
__global__ void __launch_bounds__(FORCE_THREADS) force_kernel_queue_group(ForceArgs a)
{
    PEDONI_FORCE_LDS(SLOTS);
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t block = a.xcd_remap ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    force_queue_tile_group<MODE, SLOTS, G>(a, block * FORCE_WAVES + wave, queue_all[wave], who_all[wave], tab);
}




template <int MODE, int SLOTS>
__global__ void __launch_bounds__(FORCE_THREADS) force_kernel_queue(ForceArgs a)
{
    force_queue_body<MODE, SLOTS>(a);
}

// The same kernel held to 94 SGPRs: a CU admits floor(800 / (ceil(sgpr / 16) * 16 + 16)) 256-thread
// workgroups (MI355X_MICROARCH.md, residency), i.e. 6 at the 97-112 the compiler uses by
// default and 7 at <= 96 -- the 7th wave per SIMD that the <= 72 VGPRs and a 5-slot queue
// (22 KB LDS per block) already allow.
template <int MODE, int SLOTS>
__global__ void __launch_bounds__(FORCE_THREADS, 7) __attribute__((amdgpu_num_sgpr(94)))
force_kernel_queue_s94(ForceArgs a)
{
    force_queue_body<MODE, SLOTS>(a);
}

// ---- K_FORCE of a band, edge rows first (the overlapped tick of a sharded run) --------------------
// One launch over the whole band, as the plain tick's -- but the workgroups the hardware starts first
// take the tiles of the band's edge rows (the head and the tail of the sorted order), and the wave
// that completes the last of those tiles says so in a word of device memory.  The shard's
// communication stream waits on that word (edge_wait_kernel below) and packs and sends the updated
// edge rows under the rest of this launch: no second force launch, no event between two launches on
// the model's stream.
//   hardware block b <  e_lo          -> tile b                         (the head of the sorted order)
//   e_lo <= b < e_lo + e_hi           -> tile t_hi + (b - e_lo)         (around the end of the live agents)
//   the others                        -> the tiles left, in order, XCD-contiguous as ever
// That placement is a HINT from the host (edge_blocks / edge_tile_hi: where the lists' capacity and its own
// bound of the live count put the edge rows) and a speed matter only.  Which workgroups really hold edge
// agents, and how many there are, each workgroup reads off cell_start and the live count AFTER its tile
// (off the path to its first load: read before the tile they cost every wave one more dependent latency,
// force kernel 91 against 87 us): an edge workgroup releases its records and counts itself in, the one
// that completes the count stores the flag.  One release per edge WORKGROUP, and as few of those as there
// are: at agent scope it writes back the XCD's whole L2 (126 blocks per side hinted AND signalling, one
// release per wave: force kernel 118 us).  Every launch stores the flag: with no edge agent at all,
// workgroup 0 does.
struct EdgeEpilogue {       // what the workgroup's last lanes need when they leave the tile, parked in LDS before it
    const uint32_t* lo_end;   // &cell_start[edge_row[0] * cols]
    const uint32_t* hi_begin; // &cell_start[edge_row[1] * cols]
    const uint32_t* live;
    uint32_t* counter;
    uint32_t* flag;
    uint32_t base, seq, tile, first_block, lanes_out;
};

// (see NoExit) the lanes that leave the tile together count themselves out of the workgroup; those that
// complete the count speak for it: an edge workgroup releases its records and counts itself in, the
// workgroup that completes THAT count stores the flag
struct EdgeExit {
    EdgeEpilogue* ep;
    __device__ __forceinline__ void operator()() const
    {
        const unsigned long long here = __ballot(1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // these lanes' records are out
        if (__builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u)) != 0u) return;
        const uint32_t n = (uint32_t)__popcll(here);
        if (__hip_atomic_fetch_add(&ep->lanes_out, n, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + n != (uint32_t)FORCE_THREADS) return;
        const uint32_t base = ep->base;
        const uint32_t lo_n = *ep->lo_end - base;         // agents below edge_row[0]
        const uint32_t hi_0 = *ep->hi_begin - base;       // first agent of edge_row[1]
        const uint32_t live = *ep->live - base;
        // edge workgroups: tiles [0, t_lo) and [t_h0, t_h1), the second run cut where it overlaps the first
        const uint32_t T = (uint32_t)FORCE_THREADS;
        const uint32_t t_lo = (lo_n + T - 1u) / T;
        uint32_t t_h0 = hi_0 / T, t_h1 = live > hi_0 ? (live + T - 1u) / T : t_h0;
        t_h0 = max(t_h0, t_lo); t_h1 = max(t_h1, t_h0);
        const uint32_t expected = t_lo + (t_h1 - t_h0);
        const uint32_t tile = ep->tile;
        const bool mine = expected ? (tile < t_lo || (tile >= t_h0 && tile < t_h1)) : ep->first_block != 0u;
        if (!mine) return;
        // release (agent scope): the workgroup's records are out before it counts itself in; acquire: the
        // workgroup that completes the count has every earlier one's records behind its store of the flag
        uint32_t* counter = ep->counter;
        if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == max(expected, 1u)) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
            __hip_atomic_store(ep->flag, ep->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};

template <int MODE, int SLOTS>
__device__ __forceinline__ void force_edge_first_body(const ForceArgs& a)
{
    __shared__ EdgeEpilogue ep;
    const uint32_t tile = edge_first_tile(blockIdx.x, gridDim.x, a.edge_blocks[0], a.edge_blocks[1], a.edge_tile_hi, a.xcd_remap);
    if (threadIdx.x == 0) {
        ep.lo_end = a.cell_start + (int64_t)a.edge_row[0] * a.grid.cols;
        ep.hi_begin = a.cell_start + (int64_t)a.edge_row[1] * a.grid.cols;
        ep.live = a.live_count;
        ep.counter = a.edge_counter;
        ep.flag = a.edge_flag;
        ep.base = a.base; ep.seq = a.edge_seq; ep.tile = tile; ep.first_block = blockIdx.x == 0 ? 1u : 0u;
        ep.lanes_out = 0;
    }
    PEDONI_FORCE_LDS(SLOTS);          // (its barrier publishes `ep` too)
    const uint32_t wave = threadIdx.x >> 6;
    force_queue_tile<MODE, SLOTS, NoDiag, EdgeExit>(a, tile * blockDim.x + threadIdx.x, queue_all[wave], who_all[wave], tab,
                                                          EdgeExit{&ep});
}

// The other stream's side of it: one wave that looks at the word every ~2 us (s_sleep between two agent-scope
// loads: no traffic to speak of) and ends when it reads this tick's number -- or after `limit` ticks of the
// 100 MHz clock, raising STATUS_EDGE_WAIT, so that the wave ends whatever happens to the launch it waits for.
// (hipStreamWaitValue32 does work here -- tools/microbench/stream_wait_value.hip -- but the runtime
// implements it as a kernel that polls without pause: with it in flight the place kernel took 26 us
// instead of 18.5 and the force kernel 103 instead of 92, profiles/r03_shard_timeline.txt.)
// ... and the word the NEXT tick's scan looks at (scan_rows_kernel's wait_flag): stored behind the unpack
// of that tick's lists on the communication stream
__global__ void __launch_bounds__(64) edge_post_kernel(uint32_t* flag, uint32_t seq)
{
    if (threadIdx.x != 0) return;
    __threadfence();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(64) edge_wait_kernel(const uint32_t* flag, uint32_t seq, uint32_t* status,
                                                       unsigned long long limit)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const uint32_t seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(seen - seq) >= 0) return;
        if (wall_clock64() - t0 > limit) { atomicOr(status, STATUS_EDGE_WAIT); return; }
        __builtin_amdgcn_s_sleep(64);
    }
}

template <int MODE, int SLOTS>
__global__ void __launch_bounds__(FORCE_THREADS, 7) __attribute__((amdgpu_num_sgpr(94)))
force_kernel_queue_edge_first_s94(ForceArgs a) { force_edge_first_body<MODE, SLOTS>(a); }
template <int MODE, int SLOTS>
__global__ void __launch_bounds__(FORCE_THREADS) force_kernel_queue_edge_first(ForceArgs a) { force_edge_first_body<MODE, SLOTS>(a); }


// ---- on-device periodic spawning (Simulator::tick, lib.rs:67-85 + sfm.rs:49-56) ---------------
// One thread replays, draw for draw, what the host does each tick: per periodic spawner
// count = poisson(frequency / 10) (util.rs:78-89, Knuth's product of f64 uniforms) and
// pos = p1.lerp(p2, f32()) from the position stream, then one desired speed per new agent
// from the model's own stream (Irwin-Hall(12), the build-owned generator).  A few dozen
// agents per tick: the serial kernel costs microseconds and removes the last per-tick host
// touch, so whole tick_n batches of a spawning scenario run without the host.
struct SpawnerDev {
    float x0, y0, x1, y1;
    uint32_t destination, pad;
    double exp_neg_lambda; // exp(-frequency / 10), computed by the host's libm as upstream does
};
struct SpawnState { unsigned long long rng_pos, rng_v0; };

__device__ __forceinline__ unsigned long long wyrand_next(unsigned long long& s)
{
    s += 0xa0761d6478bd642fULL;
    unsigned long long b = s ^ 0xe7037ed1a0b428dbULL;
    return __umul64hi(s, b) ^ (s * b);
}
__device__ __forceinline__ float wyrand_f32(unsigned long long& s) { return (float)(wyrand_next(s) >> 40) * 0x1.0p-24f; }
__device__ __forceinline__ double wyrand_f64(unsigned long long& s) { return (double)(wyrand_next(s) >> 11) * 0x1.0p-53; }

__global__ void spawn_kernel(const SpawnerDev* __restrict__ sp, uint32_t n_sp, SpawnState* __restrict__ st,
                             uint32_t at0, uint32_t cap, float2* __restrict__ pos,
                             float4* __restrict__ velx, uint32_t* __restrict__ dest,
                             HaloIn* __restrict__ halo)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    unsigned long long rp = st->rng_pos, rv = st->rng_v0;
    uint32_t n = 0, dropped = 0;
    for (uint32_t k = 0; k < n_sp; ++k) {
        int32_t count = 0;                                    // util.rs:78-89
        double x = wyrand_f64(rp);
        while (x >= sp[k].exp_neg_lambda) {
            x *= wyrand_f64(rp);
            count += 1;
        }
        for (int32_t c = 0; c < count; ++c) {                 // lib.rs:75-81
            float u = wyrand_f32(rp);
            float w = 1.0f - u;                               // glam lerp: a * (1 - s) + b * s
            float2 p = make_float2(sp[k].x0 * w + sp[k].x1 * u, sp[k].y0 * w + sp[k].y1 * u);
            if (n < cap) {
                pos[at0 + n] = p;
                dest[at0 + n] = sp[k].destination;
                n += 1;
            } else {
                dropped += 1;
            }
        }
    }
    for (uint32_t i = 0; i < n + dropped; ++i) {              // sfm.rs:54, one draw per agent
        float acc = 0.0f;
        for (int j = 0; j < 12; ++j) acc += wyrand_f32(rv);
        if (i < n) velx[at0 + i] = make_float4(0.0f, 0.0f, 0.0f, 1.34f + 0.26f * (acc - 6.0f)); // sfm.rs:53-54
    }
    st->rng_pos = rp;
    st->rng_v0 = rv;
    halo->n_above = n;
    halo->counted = 1;
    if (dropped) atomicOr(&halo->error, 8u);
}

// ---- halo exchange (no reference counterpart) -------------------------------
// A band owns grid rows [lo, hi).  After update_states its owned agents sit in last pass's
// sorted order, so the ones that can now be in rows {lo-1, lo} came from old rows lo, lo+1
// -- one contiguous index range read off cell_start -- and likewise {hi-1, hi} from old
// rows hi-2, hi-1.  One workgroup per direction compacts them, in index order, into a
// record list: {pos.xy, vel.xy, desired_speed, destination}.
struct HaloList {            // device layout of one direction's buffer
    uint32_t count, flags, pad0, pad1;       // PEDONI_HALO_HEADER_WORDS
    // followed by count records of PEDONI_HALO_RECORD_WORDS words
};

// (256 threads: in the overlapped sharded tick this kernel is launched while the interior rows' force
// kernel holds the chip, and a 1024-thread workgroup needs 16 free wave slots on ONE CU before it can
// start -- it waited 50 us for them and the exchange it feeds started that much later.  A 256-thread
// workgroup fits wherever one force workgroup retires.)
__global__ void __launch_bounds__(256)
halo_pack_kernel(const float2* __restrict__ pos, const float4* __restrict__ velx,
                 const uint32_t* __restrict__ dest, const uint32_t* __restrict__ cs, GridView grid, int32_t band_lo, int32_t band_hi,
                 uint32_t cap_each, uint32_t* __restrict__ send)
{
    __shared__ uint32_t lds[16];
    const int dir = blockIdx.x;                           // 0 = down (for the band below), 1 = up
    const uint32_t words_each = PEDONI_HALO_HEADER_WORDS + cap_each * PEDONI_HALO_RECORD_WORDS;
    uint32_t* out = send + (size_t)dir * words_each;
    uint32_t* rec = out + PEDONI_HALO_HEADER_WORDS;
    int32_t row_a, row_b, want_a, want_b;
    if (dir == 0) { row_a = band_lo; row_b = min(band_lo + 2, band_hi); want_a = band_lo - 1; want_b = band_lo; }
    else          { row_a = max(band_hi - 2, band_lo); row_b = band_hi; want_a = band_hi - 1; want_b = band_hi; }
    const bool has_neighbour = dir == 0 ? band_lo > 0 : band_hi < grid.rows;
    uint32_t begin = cs[(int64_t)row_a * grid.cols], end = cs[(int64_t)row_b * grid.cols];
    if (!has_neighbour) end = begin;
    uint32_t written = 0, flags = 0;
    for (uint32_t chunk = begin; chunk < end; chunk += blockDim.x) {
        uint32_t i = chunk + threadIdx.x;
        uint32_t take = 0;
        float2 p = make_float2(0.0f, 0.0f);
        if (i < end) {
            p = pos[i];
            if (p.x == p.x && p.y == p.y) {               // NaN agents are dropped by the next pass anyway
                int32_t row = f32_as_i32(p.y / grid.unit);
                take = (row == want_a || row == want_b) ? 1u : 0u;
                if (row < band_lo - 1 || row > band_hi) flags |= 2u; // left the band by > 1 row
            }
        }
        uint32_t total;
        uint32_t ex = block_exclusive_scan(take, lds, total);
        uint32_t at = written + ex;
        if (take) {
            if (at < cap_each) {
                uint32_t* r = rec + (size_t)at * PEDONI_HALO_RECORD_WORDS;
                float4 v = velx[i];
                r[0] = __float_as_uint(p.x); r[1] = __float_as_uint(p.y);
                r[2] = __float_as_uint(v.x); r[3] = __float_as_uint(v.y);
                r[4] = __float_as_uint(v.w); r[5] = dest[i];
            } else {
                flags |= 1u;                              // list overflow
            }
        }
        written += total;
    }
    // block-wide OR of the flags
    __shared__ uint32_t flag_or;
    if (threadIdx.x == 0) flag_or = 0;
    __syncthreads();
    if (flags) atomicOr(&flag_or, flags);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = min(written, cap_each);
        out[1] = flag_or;
        out[2] = 0; out[3] = 0;
    }
}

// Places the list of the band below right in front of the own agents ([base - n, base)) and
// the list of the band above behind everything stored ([gap_end, gap_end + n)): with the
// stable cell sort this reproduces the single-GPU order (lower bands hold lower indices).
__global__ void halo_unpack_kernel(const uint32_t* __restrict__ from_below,
                                   const uint32_t* __restrict__ from_above, uint32_t cap_each,
                                   uint32_t n_behind, uint32_t base, uint32_t gap_end,
                                   float2* __restrict__ pos, float4* __restrict__ velx,
                                   uint32_t* __restrict__ dest, HaloIn* __restrict__ halo,
                                   FieldView field, GridView grid, int32_t band_lo, int32_t band_hi,
                                   uint32_t parity, SortFlags* __restrict__ flags,
                                   uint32_t* __restrict__ key, uint32_t* __restrict__ cell_count,
                                   uint32_t* __restrict__ row_count)
{
    // thread t < cap: landing slot base - cap + t (the list is right-aligned against base);
    // thread t >= cap: slot gap_end + (t - cap), for the n_behind slots by which the host's
    // bound of the arrays grows this tick (the list from above fills the first of them).
    // Every slot gets a key -- the record's cell (same arithmetic as key_kernel) or DEAD for
    // an unused slot -- so the pass that follows needs no K_KEY launch for the exchanged
    // agents and never meets a stale key.
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cap_each + n_behind) return;
    uint32_t n_below = from_below ? min(from_below[0], cap_each) : 0u;
    uint32_t n_above = from_above ? min(from_above[0], cap_each) : 0u;
    if (t == 0) {
        halo->n_below = n_below;
        halo->n_above = n_above;
        halo->counted = 0;
        halo->sharded = 1;
        uint32_t err = (from_below ? from_below[1] : 0u) | (from_above ? from_above[1] : 0u);
        if (err) atomicOr(&halo->error, err);
    }
    const uint32_t* src = nullptr;
    uint32_t at;
    if (t < cap_each) {
        at = base - cap_each + t;
        uint32_t unused = cap_each - n_below;
        if (t >= unused) src = from_below + PEDONI_HALO_HEADER_WORDS + (size_t)(t - unused) * PEDONI_HALO_RECORD_WORDS;
    } else {
        uint32_t k = t - cap_each;
        at = gap_end + k;
        if (k < n_above) src = from_above + PEDONI_HALO_HEADER_WORDS + (size_t)k * PEDONI_HALO_RECORD_WORDS;
    }
    uint32_t kk = DEAD;
    int32_t cx = 0, cy = 0;
    if (src) {
        v2 p = mk(__uint_as_float(src[0]), __uint_as_float(src[1]));
        uint32_t d = src[5];
        pos[at] = make_float2(p.x, p.y);
        velx[at] = make_float4(__uint_as_float(src[2]), __uint_as_float(src[3]), 0.0f, __uint_as_float(src[4]));
        dest[at] = d;
        if (cell_xy(grid, p, cx, cy) && cy >= band_lo - 1 && cy <= band_hi && survives(field, p, d)) {
            {
                kk = (uint32_t)cy * (uint32_t)grid.cols + (uint32_t)cx;
                // exchanged agents belong in the four boundary rows (general sort form there);
                // anywhere else the whole pass must take the general form
                if (!(cy <= band_lo || cy >= band_hi - 1)) atomicOr(&flags->far[parity], 1u);
            }
        }
    }
    key[at] = kk;
    count_key(cell_count, row_count, kk != DEAD, kk, (uint32_t)cy);
}

} // namespace pedoni

#ifdef PEDONI_DIAGNOSTICS
#include "kernels_diag.hpp"
#endif
