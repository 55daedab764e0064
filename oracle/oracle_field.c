/* oracle_field.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Restatement of pedoni-simulator/src/field.rs (sampling :235-258, builder :16-192).
 */
#include "pedoni_oracle.h"
#include "oracle_math.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

/* field.rs:235-239 */
float oracle_get_potential(const oracle_field* f, uint32_t waypoint, float px, float py)
{
    float qx = px / f->unit - 0.5f, qy = py / f->unit - 0.5f;
    return oracle_bilinear(f->potential_maps[waypoint], f->rows, f->cols, qx, qy);
}

/* field.rs:242-245 */
float oracle_get_obstacle_distance(const oracle_field* f, float px, float py)
{
    float qx = px / f->unit - 0.5f, qy = py / f->unit - 0.5f;
    return oracle_bilinear(f->distance_map, f->rows, f->cols, qx, qy);
}

/* field.rs:248-252 */
void oracle_get_potential_grad(const oracle_field* f, uint32_t waypoint, float px, float py,
                               float* out_xy)
{
    float qx = px / f->unit - 0.5f, qy = py / f->unit - 0.5f;
    oracle_sobel_filter(f->potential_maps[waypoint], f->rows, f->cols, qx, qy, out_xy);
}

/* field.rs:255-258 */
void oracle_get_obstacle_distance_grad(const oracle_field* f, float px, float py, float* out_xy)
{
    float qx = px / f->unit - 0.5f, qy = py / f->unit - 0.5f;
    oracle_sobel_filter(f->distance_map, f->rows, f->cols, qx, qy, out_xy);
}

/* field.rs:25-26: grid_size = (size / unit).ceil(); shape = (y as usize, x as usize) */
void oracle_field_shape(float size_x, float size_y, float unit, int32_t* rows, int32_t* cols)
{
    *rows = (int32_t)o_f32_as_usize(ceilf(size_y / unit));
    *cols = (int32_t)o_f32_as_usize(ceilf(size_x / unit));
}

/* ---- rasterisation ------------------------------------------------------------
 * field.rs:42-64,66-88 rasterise a CLOSED LineString (the 4-vertex rectangle of
 * util::line_with_width, scaled by 1/unit; f32 coordinates, converted to f64 by the crate)
 * with geo-rasterize 0.1.2 (Cargo.lock:488-489; source not under /root/reference).
 *
 * PARITY UNPINNED.  The crate documents its rasterisation as ports of GDAL's burners --
 * filled polygons from GDALdllImageFilledPolygon, line strings from
 * GDALdllImageLineAllTouched -- checked against GDAL by property tests.  What follows
 * restates THAT published algorithm (GDAL alg/llrasterize.cpp, all-touched line burner):
 * left-to-right segments; a segment that stays in one pixel column, or is closer than 0.01
 * to vertical, burns a straight run of that column (likewise rows for near-horizontal ones);
 * anything else is walked pixel by pixel along x with the slope, moving to the next
 * scanline whenever the step to the next column would cross a row boundary, nudged by 1e-9
 * so that it always advances.  The product's builder (pedoni_amd/csrc/host/field.cpp) states the
 * SAME published walk a second time, in C++: the two agreeing bit for bit on every scenario is a
 * consistency check between two restatements of one algorithm, NOT a pin and not independent
 * evidence.  The independent statement is the exact grid traversal inside the tests
 * (tests/test_host_cpu.py::_exact_traversal); where it and this walk differ -- corner / end-point
 * ties, a handful of cells -- is committed as tests/golden/burner_corner_ties.json, the first
 * place to look once geo-rasterize's own output exists (the reference's test of the crate only
 * prints, field.rs:272-286).  Affects how the INPUT maps are produced, not the per-step arithmetic.
 */
static void burn(uint8_t* mask, int32_t rows, int32_t cols, int64_t c, int64_t r)
{
    if (c >= 0 && r >= 0 && c < cols && r < rows) mask[(size_t)r * cols + c] = 1;
}

static void swap_d(double* a, double* b) { double t = *a; *a = *b; *b = t; }

static void burn_segment(uint8_t* mask, int32_t rows, int32_t cols,
                         double x, double y, double x_end, double y_end)
{
    const double w = (double)cols, h = (double)rows;
    /* segments wholly off the raster */
    if ((y < 0 && y_end < 0) || (y > h && y_end > h) || (x < 0 && x_end < 0) || (x > w && x_end > w)) return;
    if (x > x_end) { swap_d(&x, &x_end); swap_d(&y, &y_end); }       /* proceed left to right */

    if (floor(x) == floor(x_end) || fabs(x - x_end) < 0.01) {         /* (near) vertical */
        if (y_end < y) swap_d(&y, &y_end);
        const int64_t ix = (int64_t)floor(x_end);
        int64_t iy = (int64_t)floor(y), iy_end = (int64_t)floor(y_end);
        if (ix < 0 || ix >= cols) return;
        if (iy < 0) iy = 0;
        if (iy_end >= rows) iy_end = rows - 1;
        for (; iy <= iy_end; ++iy) burn(mask, rows, cols, ix, iy);
        return;
    }
    if (floor(y) == floor(y_end) || fabs(y - y_end) < 0.01) {         /* (near) horizontal */
        int64_t ix = (int64_t)floor(x), ix_end = (int64_t)floor(x_end);
        const int64_t iy = (int64_t)floor(y);
        if (iy < 0 || iy >= rows) return;
        if (ix < 0) ix = 0;
        if (ix_end >= cols) ix_end = cols - 1;
        for (; ix <= ix_end; ++ix) burn(mask, rows, cols, ix, iy);
        return;
    }

    const double slope = (y_end - y) / (x_end - x);
    if (x_end > w) { y_end -= (x_end - w) * slope; x_end = w; }       /* clip in x */
    if (x < 0.0) { y += (0.0 - x) * slope; x = 0.0; }
    if (y_end > y) {                                                   /* clip in y */
        if (y < 0.0) { x += (0.0 - y) / slope; y = 0.0; }
        if (y_end >= h) x_end += (y_end - h) / slope;
    } else {
        if (y >= h) { x += (h - y) / slope; y = h; }
        if (y_end < 0.0) x_end -= (y_end - 0.0) / slope;
    }
    while (x >= 0.0 && x < x_end) {                                    /* pixel to pixel */
        const int64_t ix = (int64_t)floor(x), iy = (int64_t)floor(y);
        if (iy >= 0 && iy < rows) burn(mask, rows, cols, ix, iy);
        double step_x = floor(x + 1.0) - x;
        double step_y = step_x * slope;
        if ((int64_t)floor(y + step_y) == iy) {                        /* next column, same scanline */
            x += step_x; y += step_y;
        } else if (slope < 0) {
            step_y = (double)iy - y;
            if (step_y > -0.000000001) step_y = -0.000000001;
            step_x = step_y / slope;
            x += step_x; y += step_y;
        } else {
            step_y = (double)(iy + 1) - y;
            if (step_y < 0.000000001) step_y = 0.000000001;
            step_x = step_y / slope;
            x += step_x; y += step_y;
        }
    }
}

void oracle_rasterize_outline(const float* v, int32_t rows, int32_t cols, uint8_t* mask)
{
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3; /* shape.close(): last vertex joins the first */
        burn_segment(mask, rows, cols, v[2 * i], v[2 * i + 1], v[2 * j], v[2 * j + 1]);
    }
}

/* ---- fast marching ------------------------------------------------------------
 * field.rs:118-192.  BinaryHeap<(Reverse<NotNan<f32>>, Index)> is a max-heap on the
 * tuple: smallest u first; among equal u the LARGEST Index {y, x} (derive(Ord), y
 * major) first.  That total order makes the pop sequence independent of the heap
 * implementation, so any correct heap reproduces it.
 */
typedef struct { float u; int32_t y, x; } fmm_item;
typedef struct { fmm_item* a; size_t n, cap; } fmm_heap;

/* returns 1 if p should pop before q */
static inline int fmm_before(const fmm_item* p, const fmm_item* q)
{
    if (p->u != q->u) return p->u < q->u;
    if (p->y != q->y) return p->y > q->y;
    return p->x > q->x;
}

static void heap_push(fmm_heap* h, fmm_item it)
{
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 1024;
        h->a = (fmm_item*)realloc(h->a, h->cap * sizeof(fmm_item));
    }
    size_t i = h->n++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!fmm_before(&it, &h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = it;
}

static fmm_item heap_pop(fmm_heap* h)
{
    fmm_item top = h->a[0];
    fmm_item last = h->a[--h->n];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m;
        if (l >= h->n) break;
        m = (r < h->n && fmm_before(&h->a[r], &h->a[l])) ? r : l;
        if (!fmm_before(&h->a[m], &last)) break;
        h->a[i] = h->a[m];
        i = m;
    }
    if (h->n) h->a[i] = last;
    return top;
}

static inline int in_grid(int32_t rows, int32_t cols, int64_t x, int64_t y)
{
    return x >= 0 && y >= 0 && y < rows && x < cols;
}

/* potential.get(ix).cloned().unwrap_or(f32::MAX) */
static inline float pot_or_max(const float* p, int32_t rows, int32_t cols, int64_t x, int64_t y)
{
    return in_grid(rows, cols, x, y) ? p[(size_t)y * cols + x] : FLT_MAX;
}

void oracle_apply_fmm(float* potential, const float* f, int32_t rows, int32_t cols)
{
    /* the (j, i) pairs of field.rs:134,156: ix.add(i, j) -> x += i, y += j */
    static const int DJ[4] = { -1, 1, 0, 0 };
    static const int DI[4] = { 0, 0, -1, 1 };
    uint8_t* accepted = (uint8_t*)calloc((size_t)rows * cols, 1);
    fmm_heap heap = { 0, 0, 0 };

    for (int32_t y = 0; y < rows; ++y) {          /* :128-146 */
        for (int32_t x = 0; x < cols; ++x) {
            size_t ix = (size_t)y * cols + x;
            if (potential[ix] == 0.0f) {
                accepted[ix] = 1;
                for (int k = 0; k < 4; ++k) {
                    int64_t nx = (int64_t)x + DI[k], ny = (int64_t)y + DJ[k];
                    if (!in_grid(rows, cols, nx, ny)) continue;
                    size_t nix = (size_t)ny * cols + nx;
                    if (potential[nix] != 0.0f) {
                        float u = f[nix];
                        potential[nix] = u;
                        fmm_item it = { u, (int32_t)ny, (int32_t)nx };
                        heap_push(&heap, it);
                    }
                }
            }
        }
    }

    while (heap.n) {                              /* :148-191 */
        fmm_item top = heap_pop(&heap);
        size_t ix = (size_t)top.y * cols + top.x;
        if (accepted[ix]) continue;
        accepted[ix] = 1;
        float u = top.u;

        for (int k = 0; k < 4; ++k) {
            int j = DJ[k], i = DI[k];
            int64_t nx = (int64_t)top.x + i, ny = (int64_t)top.y + j;
            if (!in_grid(rows, cols, nx, ny)) continue;       /* None */
            size_t nix = (size_t)ny * cols + nx;
            if (accepted[nix]) continue;                       /* Some(true) */

            float fv = f[nix];
            float u1, u2;
            if (j == 0) {                          /* :163-166 */
                float u2a = pot_or_max(potential, rows, cols, nx, ny - 1);
                float u2b = pot_or_max(potential, rows, cols, nx, ny + 1);
                u1 = u;
                u2 = fminf(u2a, u2b);
            } else {                               /* :167-171 */
                float u1a = pot_or_max(potential, rows, cols, nx - 1, ny);
                float u1b = pot_or_max(potential, rows, cols, nx + 1, ny);
                u1 = fminf(u1a, u1b);
                u2 = u;
            }

            float un;
            if (u1 == FLT_MAX) {
                un = u2 + fv;
            } else if (u2 == FLT_MAX) {
                un = u1 + fv;
            } else {
                float d = u1 - u2;
                float sq = 2.0f * fv * fv - d * d;             /* :178 */
                if (sq >= 0.0f) un = (u1 + u2 + sqrtf(sq)) / 2.0f;
                else            un = fminf(u1, u2) + fv;
            }

            if (un < potential[nix]) {
                potential[nix] = un;
                fmm_item it = { un, (int32_t)ny, (int32_t)nx };
                heap_push(&heap, it);
            }
        }
    }
    free(heap.a);
    free(accepted);
}

/* field.rs:16-114,220-232: FieldBuilder::new / add_obstacle / add_waypoint / build */
void oracle_field_build(float size_x, float size_y, float unit,
                        const oracle_segment* obstacles, uint32_t n_obstacles,
                        const oracle_segment* waypoints, uint32_t n_waypoints,
                        uint8_t* obstacle_exist, float* distance_map, float* potential_maps)
{
    int32_t rows, cols;
    oracle_field_shape(size_x, size_y, unit, &rows, &cols);
    size_t n = (size_t)rows * cols;

    memset(obstacle_exist, 0, n);                 /* :27 */
    for (int32_t x = 0; x < cols; ++x) {          /* :29-30 */
        obstacle_exist[x] = 1;
        obstacle_exist[(size_t)(rows - 1) * cols + x] = 1;
    }
    for (int32_t y = 0; y < rows; ++y) {          /* :31-32 */
        obstacle_exist[(size_t)y * cols] = 1;
        obstacle_exist[(size_t)y * cols + cols - 1] = 1;
    }

    float verts[8], line[4];
    for (uint32_t o = 0; o < n_obstacles; ++o) {  /* :42-64 */
        line[0] = obstacles[o].x0; line[1] = obstacles[o].y0;
        line[2] = obstacles[o].x1; line[3] = obstacles[o].y1;
        oracle_line_with_width(line, obstacles[o].width, verts);
        for (int k = 0; k < 8; ++k) verts[k] = verts[k] / unit;
        oracle_rasterize_outline(verts, rows, cols, obstacle_exist); /* `*a |= b` */
    }

    uint8_t* mask = (uint8_t*)malloc(n);
    for (uint32_t w = 0; w < n_waypoints; ++w) {  /* :66-88: background f32::MAX, label 0 */
        float* pm = potential_maps + (size_t)w * n;
        line[0] = waypoints[w].x0; line[1] = waypoints[w].y0;
        line[2] = waypoints[w].x1; line[3] = waypoints[w].y1;
        oracle_line_with_width(line, waypoints[w].width, verts);
        for (int k = 0; k < 8; ++k) verts[k] = verts[k] / unit;
        memset(mask, 0, n);
        oracle_rasterize_outline(verts, rows, cols, mask);
        for (size_t i = 0; i < n; ++i) pm[i] = mask[i] ? 0.0f : FLT_MAX;
    }
    free(mask);

    float* slow = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) {              /* :98-99 */
        distance_map[i] = obstacle_exist[i] ? 0.0f : 1e24f;
        slow[i] = unit;
    }
    oracle_apply_fmm(distance_map, slow, rows, cols);

    for (size_t i = 0; i < n; ++i)                /* :102 */
        slow[i] = unit * (obstacle_exist[i] ? 1e6f : 1.0f);
    for (uint32_t w = 0; w < n_waypoints; ++w)    /* :103-105 (rayon; order-free) */
        oracle_apply_fmm(potential_maps + (size_t)w * n, slow, rows, cols);
    free(slow);
}
