// eikonal.hpp -- opt-in GPU builder of the field maps: a parallel solver
// of the eikonal equation |grad u| = f that upstream's Field::from_scenario solves with a heap
// fast-marching pass (field.rs:118-192).  Included at the end of pedoni_hip.hip.
//
// NOT a parity path, by construction.  Upstream's pass updates a cell from its neighbours'
// TENTATIVE values and accepts whenever `2 f^2 - (u1 - u2)^2 >= 0`, so its numbers depend on
// the heap's pop order (tools/fmm_order_dependence.py); no parallel solver reproduces them.
// What this solver computes is the unique fixed point of the standard first-order upwind
// (Godunov) update on the same 4-neighbour grid with the same per-cell slowness:
//     a = min(left, right), b = min(up, down)
//     u = min(a, b) + f                      if |a - b| >= f
//         (a + b + sqrt(2 f^2 - (a - b)^2)) / 2   otherwise
// -- the solution upstream's pass approximates.  The bit-parity tests of the per-step path
// never use it; `Field::build(..., solver="gpu")` is for start-up time on large fields
// (a 4000 x 4000 map: ~1 s on the host heap, tens of ms here).
//
// Method: block-based fast iterative method.  The grid is cut into 16 x 16 tiles; an active
// tile is relaxed in LDS (in-place sweeps, monotone: values only decrease, so any order
// converges to the same fixed point) and, if anything in it changed, re-activates itself and
// its four neighbours.  The host relaunches over the tile map until no tile is active.
#pragma once

namespace {

constexpr int EIK_TILE = 16;
constexpr int EIK_SWEEPS = 24;

__device__ __forceinline__ float eik_update(float a, float b, float f)
{
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    if (!(lo < 1e23f)) return lo;                      // no finite neighbour yet
    if (!(hi < 1e23f) || hi - lo >= f) return lo + f;
    const float d = a - b;
    return (a + b + sqrtf(2.0f * f * f - d * d)) * 0.5f;
}

__global__ void __launch_bounds__(EIK_TILE* EIK_TILE)
eikonal_tile_kernel(float* __restrict__ u, const float* __restrict__ slowness, float uniform_f, int32_t rows,
                    int32_t cols, int32_t tiles_x, int32_t tiles_y, const uint8_t* __restrict__ active_cur,
                    uint8_t* __restrict__ active_next, uint32_t* __restrict__ n_active_next)
{
    const int32_t tile = (int32_t)blockIdx.x;
    if (!active_cur[tile]) return;
    const int32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lx = threadIdx.x % EIK_TILE, ly = threadIdx.x / EIK_TILE;
    const int32_t gx = tx * EIK_TILE + lx, gy = ty * EIK_TILE + ly;
    __shared__ float t[EIK_TILE + 2][EIK_TILE + 2];
    __shared__ int changed_any;
    const float INF = 3.0e38f;
    auto at = [&](int32_t y, int32_t x) -> float {
        return (x >= 0 && y >= 0 && x < cols && y < rows) ? u[(size_t)y * (size_t)cols + (size_t)x] : INF;
    };
    const bool inside = gx < cols && gy < rows;
    const float mine0 = inside ? at(gy, gx) : INF;
    t[ly + 1][lx + 1] = mine0;
    if (ly == 0) t[0][lx + 1] = at(gy - 1, gx);
    if (ly == EIK_TILE - 1) t[EIK_TILE + 1][lx + 1] = at(gy + 1, gx);
    if (lx == 0) t[ly + 1][0] = at(gy, gx - 1);
    if (lx == EIK_TILE - 1) t[ly + 1][EIK_TILE + 1] = at(gy, gx + 1);
    if (threadIdx.x == 0) changed_any = 0;
    const float f = slowness ? (inside ? slowness[(size_t)gy * (size_t)cols + (size_t)gx] : 1.0f) : uniform_f;
    const bool source = mine0 == 0.0f;                  // zero set: fixed
    __syncthreads();
    float mine = mine0;
    for (int s = 0; s < EIK_SWEEPS; ++s) {
        float cand = mine;
        if (inside && !source) {
            const float a = fminf(t[ly + 1][lx], t[ly + 1][lx + 2]);
            const float b = fminf(t[ly][lx + 1], t[ly + 2][lx + 1]);
            cand = fminf(mine, eik_update(a, b, f));
        }
        __syncthreads();
        if (cand < mine) {
            mine = cand;
            t[ly + 1][lx + 1] = cand;
        }
        __syncthreads();
    }
    if (mine < mine0) {
        u[(size_t)gy * (size_t)cols + (size_t)gx] = mine;
        changed_any = 1;                                 // benign race: every writer stores 1
    }
    __syncthreads();
    if (threadIdx.x == 0 && changed_any) {
        active_next[tile] = 1;
        if (tx > 0) active_next[tile - 1] = 1;
        if (tx + 1 < tiles_x) active_next[tile + 1] = 1;
        if (ty > 0) active_next[tile - tiles_x] = 1;
        if (ty + 1 < tiles_y) active_next[tile + tiles_x] = 1;
        atomicAdd(n_active_next, 1u);
    }
}

} // namespace

extern "C" int pedoni_hip_eikonal(int device, float* potential, const float* slowness, float uniform_slowness,
                                  uint32_t rows, uint32_t cols, uint32_t* launches_out)
{
    if (!potential || rows == 0 || cols == 0 || (uint64_t)rows * cols > 0x7fffffffull ||
        (!slowness && !(uniform_slowness > 0.0f)))
        return fail(PEDONI_E_INVALID, "eikonal: bad arguments");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(PEDONI_E_NO_DEVICE, "eikonal: no HIP device (this builder has no CPU fallback; "
                                        "the host builder Field::from_scenario is the reference-order one)");
    if (device < 0 || device >= n_dev) return fail(PEDONI_E_INVALID, "eikonal: bad device index");
    HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)rows * cols;
    const int32_t tiles_x = ((int32_t)cols + EIK_TILE - 1) / EIK_TILE, tiles_y = ((int32_t)rows + EIK_TILE - 1) / EIK_TILE;
    const size_t n_tiles = (size_t)tiles_x * tiles_y;
    float *d_u = nullptr, *d_f = nullptr;
    uint8_t* d_active[2] = {nullptr, nullptr};
    uint32_t* d_count = nullptr;
    int rc = PEDONI_OK;
    auto cleanup = [&]() { hipFree(d_u); hipFree(d_f); hipFree(d_active[0]); hipFree(d_active[1]); hipFree(d_count); };
    if (hipMalloc((void**)&d_u, n * sizeof(float)) != hipSuccess ||
        (slowness && hipMalloc((void**)&d_f, n * sizeof(float)) != hipSuccess) ||
        hipMalloc((void**)&d_active[0], n_tiles) != hipSuccess || hipMalloc((void**)&d_active[1], n_tiles) != hipSuccess ||
        hipMalloc((void**)&d_count, 64 * sizeof(uint32_t)) != hipSuccess ||
        hipMemcpy(d_u, potential, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        (slowness && hipMemcpy(d_f, slowness, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemset(d_active[0], 1, n_tiles) != hipSuccess || hipMemset(d_active[1], 0, n_tiles) != hipSuccess) {
        cleanup();
        return fail(PEDONI_E_HIP, "eikonal: device allocation / copy failed");
    }
    // batches of BATCH relaxation launches; slot k of d_count counts the tiles launch k re-activated
    constexpr int BATCH = 32;
    uint32_t launches = 0, h_count[BATCH];
    int cur = 0;
    bool done = false;
    const uint32_t max_launches = 64u * (uint32_t)(tiles_x + tiles_y) + 1024u;   // a front crosses a tile per launch
    while (!done && launches < max_launches && rc == PEDONI_OK) {
        if (hipMemsetAsync(d_count, 0, BATCH * sizeof(uint32_t), 0) != hipSuccess) { rc = fail(PEDONI_E_HIP, "eikonal: memset failed"); break; }
        for (int k = 0; k < BATCH; ++k) {
            hipLaunchKernelGGL(eikonal_tile_kernel, dim3((uint32_t)n_tiles), dim3(EIK_TILE * EIK_TILE), 0, 0, d_u, d_f,
                               uniform_slowness, (int32_t)rows, (int32_t)cols, tiles_x, tiles_y, d_active[cur],
                               d_active[1 - cur], d_count + k);
            hipMemsetAsync(d_active[cur], 0, n_tiles, 0);          // becomes the next launch's "next" map
            cur = 1 - cur;
            launches += 1;
        }
        if (hipGetLastError() != hipSuccess ||
            hipMemcpy(h_count, d_count, sizeof h_count, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(PEDONI_E_HIP, "eikonal: launch / copy failed");
            break;
        }
        for (int k = 0; k < BATCH; ++k)
            if (h_count[k] == 0) { done = true; break; }          // a launch that changed nothing: fixed point
    }
    if (rc == PEDONI_OK && !done) rc = fail(PEDONI_E_HIP, "eikonal: no fixed point within the launch budget");
    if (rc == PEDONI_OK && hipMemcpy(potential, d_u, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(PEDONI_E_HIP, "eikonal: copy back failed");
    cleanup();
    if (launches_out) *launches_out = launches;
    return rc;
}
