// field.cpp -- Field::from_scenario and field sampling (pedoni-simulator/src/field.rs),
// util::bilinear / sobel_filter / line_with_width (util.rs).
//
// Host-side, one-off input producer of the hot path.  The fast-marching pass keeps the
// reference's pop order -- BinaryHeap<(Reverse<NotNan<f32>>, Index)>: smallest value
// first, ties broken towards the LARGEST (y, x) -- by packing (value bits, ~linear index)
// into one 64-bit key of a min-heap: potentials are non-negative, so their IEEE bit
// patterns order like the values.  Potential maps are built in parallel threads where
// the reference uses rayon (field.rs:103-105).
#include "pedoni_host.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <queue>
#include <stdexcept>
#include <thread>

namespace pedoni_host {
namespace {

inline int32_t f32_as_i32(float v) // Rust `as i32`
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

inline size_t f32_as_usize(float v) // Rust `as usize`
{
    if (!(v > 0.0f)) return 0;
    if (v >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)v;
}

inline float texel(const std::vector<float>& g, size_t rows, size_t cols, int64_t x, int64_t y)
{
    if (x < 0 || y < 0 || (uint64_t)y >= rows || (uint64_t)x >= cols) return 1e12f; // util.rs:45
    return g[(size_t)y * cols + (size_t)x];
}

inline uint32_t bits_of(float v)
{
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return u;
}
inline float float_of(uint32_t u)
{
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

// Line burner of a closed LineString (field.rs:42-64,66-88 -> geo-rasterize 0.1.2).
// PARITY UNPINNED: the crate's source is not under /root/reference and upstream's only test
// of it prints (field.rs:272-286).  The crate documents its line rasterisation as a port of
// GDAL's all-touched burner (alg/llrasterize.cpp, GDALdllImageLineAllTouched), so that
// published algorithm is what is followed here: runs of one pixel column / row for segments
// within one column / row or closer than 0.01 to vertical / horizontal, otherwise a
// left-to-right walk in which each step either enters the next pixel column on the same
// scanline or moves to the scanline boundary (1e-9 nudge).  Differs from an exact grid
// traversal only where a segment passes through a pixel corner or ends on a pixel edge
// (over the reference's 14 scenario files + the 3 fixtures: 3 distinct cells of the obstacle masks and
// 16 cells of waypoint outlines, of ~250 000 burnt cells -- listed in tests/golden/burner_corner_ties.json;
// tests/test_host_cpu.py::test_line_burner_is_all_touched_up_to_corner_ties).
struct Burner {
    uint8_t* mask;
    int64_t rows, cols;
    void set(int64_t c, int64_t r) const
    {
        if (c >= 0 && r >= 0 && c < cols && r < rows) mask[(size_t)r * cols + c] = 1;
    }
    void column_run(int64_t c, double ya, double yb) const
    {
        if (c < 0 || c >= cols) return;
        const int64_t r0 = std::max<int64_t>((int64_t)std::floor(std::min(ya, yb)), 0);
        const int64_t r1 = std::min<int64_t>((int64_t)std::floor(std::max(ya, yb)), rows - 1);
        for (int64_t r = r0; r <= r1; ++r) set(c, r);
    }
    void row_run(int64_t r, double xa, double xb) const
    {
        if (r < 0 || r >= rows) return;
        const int64_t c0 = std::max<int64_t>((int64_t)std::floor(xa), 0);
        const int64_t c1 = std::min<int64_t>((int64_t)std::floor(xb), cols - 1);
        for (int64_t c = c0; c <= c1; ++c) set(c, r);
    }
    void segment(double ax, double ay, double bx, double by) const
    {
        const double W = (double)cols, H = (double)rows;
        if ((ay < 0 && by < 0) || (ay > H && by > H) || (ax < 0 && bx < 0) || (ax > W && bx > W)) return;
        if (ax > bx) { std::swap(ax, bx); std::swap(ay, by); }            // a is the left end
        if (std::floor(ax) == std::floor(bx) || std::fabs(ax - bx) < 0.01)
            return column_run((int64_t)std::floor(bx), ay, by);
        if (std::floor(ay) == std::floor(by) || std::fabs(ay - by) < 0.01)
            return row_run((int64_t)std::floor(ay), ax, bx);
        const double m = (by - ay) / (bx - ax);
        // clip to the raster: x first, then the end that leaves in y
        if (bx > W) { by -= (bx - W) * m; bx = W; }
        if (ax < 0.0) { ay += (0.0 - ax) * m; ax = 0.0; }
        if (by > ay) {
            if (ay < 0.0) { ax += (0.0 - ay) / m; ay = 0.0; }
            if (by >= H) bx += (by - H) / m;
        } else {
            if (ay >= H) { ax += (H - ay) / m; ay = H; }
            if (by < 0.0) bx -= (by - 0.0) / m;
        }
        double x = ax, y = ay;
        while (x >= 0.0 && x < bx) {
            const int64_t c = (int64_t)std::floor(x), r = (int64_t)std::floor(y);
            if (r >= 0 && r < rows) set(c, r);
            double sx = std::floor(x + 1.0) - x, sy = sx * m;            // to the next pixel column
            if ((int64_t)std::floor(y + sy) != r) {                        // a scanline boundary comes first
                sy = m < 0 ? std::min((double)r - y, -0.000000001) : std::max((double)(r + 1) - y, 0.000000001);
                sx = sy / m;
            }
            x += sx;
            y += sy;
        }
    }
    // closed LineString through the 4 vertices (field.rs:44-53 `shape.close()`)
    void outline(const std::vector<Vec2>& v, float unit) const
    {
        for (size_t i = 0; i < v.size(); ++i) {
            const Vec2 a = v[i], b = v[(i + 1) % v.size()];
            segment(a.x / unit, a.y / unit, b.x / unit, b.y / unit);
        }
    }
};

} // namespace

namespace util {

float bilinear(const std::vector<float>& g, size_t rows, size_t cols, Vec2 pos) // util.rs:44-58
{
    const float bx = std::floor(pos.x), by = std::floor(pos.y);
    const float tx = pos.x - bx, ty = pos.y - by;
    const float sx = 1.0f - tx, sy = 1.0f - ty;
    const int64_t ix = f32_as_i32(bx), iy = f32_as_i32(by);
    float y = 0.0f;
    y += sy * sx * texel(g, rows, cols, ix, iy);
    y += sy * tx * texel(g, rows, cols, ix + 1, iy);
    y += ty * sx * texel(g, rows, cols, ix, iy + 1);
    y += ty * tx * texel(g, rows, cols, ix + 1, iy + 1);
    return y;
}

Vec2 sobel_filter(const std::vector<float>& g, size_t rows, size_t cols, Vec2 p) // util.rs:61-75
{
    auto at = [&](float dx, float dy) { return bilinear(g, rows, cols, Vec2{p.x + dx, p.y + dy}); };
    const float u00 = at(-1.0f, -1.0f), u01 = at(0.0f, -1.0f), u02 = at(1.0f, -1.0f);
    const float u10 = at(-1.0f, 0.0f), u12 = at(1.0f, 0.0f);
    const float u20 = at(-1.0f, 1.0f), u21 = at(0.0f, 1.0f), u22 = at(1.0f, 1.0f);
    return Vec2{u00 + u10 + u10 + u20 - u02 - u12 - u12 - u22,
                u00 + u01 + u01 + u02 - u20 - u21 - u21 - u22};
}

std::vector<Vec2> line_with_width(const Vec2 line[2], float width) // util.rs:106-111
{
    const float dx = line[1].x - line[0].x, dy = line[1].y - line[0].y;
    const float rcp = 1.0f / std::sqrt(dx * dx + dy * dy); // glam normalize
    const float ax = dx * rcp, ay = dy * rcp;
    const float bx = ay * 0.5f * width, by = -ax * 0.5f * width;
    return {Vec2{line[0].x - bx, line[0].y - by}, Vec2{line[0].x + bx, line[0].y + by},
            Vec2{line[1].x + bx, line[1].y + by}, Vec2{line[1].x - bx, line[1].y - by}};
}

} // namespace util

void apply_fmm(std::vector<float>& pot, const std::vector<float>& f, size_t rows, size_t cols)
{
    const size_t n = rows * cols;
    std::vector<uint8_t> accepted(n, 0);
    // min-heap on (value bits << 32 | ~index): smallest value, then largest index
    std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> heap;
    auto push = [&](float u, size_t ix) {
        heap.push(((uint64_t)bits_of(u) << 32) | (uint64_t)(0xffffffffu - (uint32_t)ix));
    };
    auto value_or_max = [&](int64_t x, int64_t y) {
        return (x < 0 || y < 0 || (size_t)y >= rows || (size_t)x >= cols) ? FLT_MAX
                                                                           : pot[(size_t)y * cols + x];
    };
    static const int DY[4] = {-1, 1, 0, 0}, DX[4] = {0, 0, -1, 1}; // field.rs:134,156

    for (size_t y = 0; y < rows; ++y)               // field.rs:128-146
        for (size_t x = 0; x < cols; ++x) {
            const size_t ix = y * cols + x;
            if (pot[ix] != 0.0f) continue;
            accepted[ix] = 1;
            for (int k = 0; k < 4; ++k) {
                const int64_t nx = (int64_t)x + DX[k], ny = (int64_t)y + DY[k];
                if (nx < 0 || ny < 0 || (size_t)ny >= rows || (size_t)nx >= cols) continue;
                const size_t nix = (size_t)ny * cols + nx;
                if (pot[nix] != 0.0f) {
                    pot[nix] = f[nix];
                    push(f[nix], nix);
                }
            }
        }

    while (!heap.empty()) {                         // field.rs:148-191
        const uint64_t top = heap.top();
        heap.pop();
        const size_t ix = 0xffffffffu - (uint32_t)top;
        if (accepted[ix]) continue;
        accepted[ix] = 1;
        const float u = float_of((uint32_t)(top >> 32));
        const int64_t x = (int64_t)(ix % cols), y = (int64_t)(ix / cols);
        for (int k = 0; k < 4; ++k) {
            const int64_t nx = x + DX[k], ny = y + DY[k];
            if (nx < 0 || ny < 0 || (size_t)ny >= rows || (size_t)nx >= cols) continue;
            const size_t nix = (size_t)ny * cols + nx;
            if (accepted[nix]) continue;
            const float fv = f[nix];
            float u1, u2;
            if (DY[k] == 0) { u1 = u; u2 = std::fmin(value_or_max(nx, ny - 1), value_or_max(nx, ny + 1)); }
            else            { u1 = std::fmin(value_or_max(nx - 1, ny), value_or_max(nx + 1, ny)); u2 = u; }
            float un;
            if (u1 == FLT_MAX) un = u2 + fv;
            else if (u2 == FLT_MAX) un = u1 + fv;
            else {
                const float d = u1 - u2;
                const float sq = 2.0f * fv * fv - d * d;
                un = sq >= 0.0f ? (u1 + u2 + std::sqrt(sq)) / 2.0f : std::fmin(u1, u2) + fv;
            }
            if (un < pot[nix]) {
                pot[nix] = un;
                push(un, nix);
            }
        }
    }
}

Field Field::from_scenario(const Scenario& scenario, float unit) { return build(scenario, unit, false, 0, nullptr); }

// Opt-in, NOT the reference's numbers: the same rasterisation, then the maps by the parallel
// eikonal solver of libpedoni_hip (pedoni_amd/csrc/eikonal.hpp) instead of the heap pass.
Field Field::from_scenario_gpu(const Scenario& scenario, float unit, int device, uint32_t* launches)
{
    return build(scenario, unit, true, device, launches);
}

Field Field::build(const Scenario& scenario, float unit, bool gpu, int device, uint32_t* launches)
{
    if (!(unit > 0.0f)) throw std::runtime_error("Field::from_scenario: unit must be > 0");
    Field fld;
    fld.unit = unit;
    fld.rows = f32_as_usize(std::ceil(scenario.field.size.y / unit)); // field.rs:25-26
    fld.cols = f32_as_usize(std::ceil(scenario.field.size.x / unit));
    if (fld.rows == 0 || fld.cols == 0 || fld.rows > 0x7fffffffu || fld.cols > 0x7fffffffu ||
        fld.rows * fld.cols > 0xfffffff0ull)
        throw std::runtime_error("Field::from_scenario: field shape out of range");
    const size_t rows = fld.rows, cols = fld.cols, n = rows * cols;

    fld.obstacle_exist.assign(n, 0);                          // field.rs:27-32
    for (size_t x = 0; x < cols; ++x) fld.obstacle_exist[x] = fld.obstacle_exist[(rows - 1) * cols + x] = 1;
    for (size_t y = 0; y < rows; ++y) fld.obstacle_exist[y * cols] = fld.obstacle_exist[y * cols + cols - 1] = 1;

    Burner ob{fld.obstacle_exist.data(), (int64_t)rows, (int64_t)cols};
    for (const ObstacleConfig& o : scenario.obstacles)        // field.rs:42-64
        ob.outline(util::line_with_width(o.line, o.width), unit);

    fld.potential_maps.resize(scenario.waypoints.size());
    std::vector<uint8_t> mask(n);
    for (size_t w = 0; w < scenario.waypoints.size(); ++w) {  // field.rs:66-88
        std::fill(mask.begin(), mask.end(), 0);
        Burner wb{mask.data(), (int64_t)rows, (int64_t)cols};
        wb.outline(util::line_with_width(scenario.waypoints[w].line, scenario.waypoints[w].width), unit);
        auto& pm = fld.potential_maps[w];
        pm.resize(n);
        for (size_t i = 0; i < n; ++i) pm[i] = mask[i] ? 0.0f : FLT_MAX;
    }

    // field.rs:98-105: distance map (slowness = unit) and one potential map per waypoint
    // (slowness = unit, x 1e6 on obstacle cells), all independent -> one thread each
    fld.distance_map.resize(n);
    std::vector<float> slow_free(n, unit), slow_obs(n);
    for (size_t i = 0; i < n; ++i) {
        fld.distance_map[i] = fld.obstacle_exist[i] ? 0.0f : 1e24f;
        slow_obs[i] = unit * (fld.obstacle_exist[i] ? 1e6f : 1.0f);
    }
    if (gpu) {
        uint32_t total = 0, k = 0;
        auto solve = [&](std::vector<float>& u, const float* slowness, float uniform) {
            if (pedoni_hip_eikonal(device, u.data(), slowness, uniform, (uint32_t)rows, (uint32_t)cols, &k) != PEDONI_OK)
                throw std::runtime_error(std::string("Field::from_scenario_gpu: ") + pedoni_hip_last_error());
            total += k;
        };
        solve(fld.distance_map, nullptr, unit);
        for (auto& pm : fld.potential_maps) solve(pm, slow_obs.data(), 0.0f);
        if (launches) *launches = total;
        return fld;
    }
    std::vector<std::thread> workers;
    workers.emplace_back([&] { apply_fmm(fld.distance_map, slow_free, rows, cols); });
    for (auto& pm : fld.potential_maps)
        workers.emplace_back([&pm, &slow_obs, rows, cols] { apply_fmm(pm, slow_obs, rows, cols); });
    for (auto& t : workers) t.join();
    return fld;
}

float Field::get_potential(size_t waypoint_id, Vec2 p) const
{
    return util::bilinear(potential_maps.at(waypoint_id), rows, cols, Vec2{p.x / unit - 0.5f, p.y / unit - 0.5f});
}
float Field::get_obstacle_distance(Vec2 p) const
{
    return util::bilinear(distance_map, rows, cols, Vec2{p.x / unit - 0.5f, p.y / unit - 0.5f});
}
Vec2 Field::get_potential_grad(size_t waypoint_id, Vec2 p) const
{
    return util::sobel_filter(potential_maps.at(waypoint_id), rows, cols, Vec2{p.x / unit - 0.5f, p.y / unit - 0.5f});
}
Vec2 Field::get_obstacle_distance_grad(Vec2 p) const
{
    return util::sobel_filter(distance_map, rows, cols, Vec2{p.x / unit - 0.5f, p.y / unit - 0.5f});
}

} // namespace pedoni_host
