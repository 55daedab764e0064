/* oracle_util.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Restatement of pedoni-simulator/src/util.rs plus the build-owned RNG.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include "pedoni_oracle.h"
#include "oracle_math.h"

#include <string.h>

/* util.rs:30-36 `Index::index_checked`: negative -> None, else usize bounds check.
 * `ix.add(..)` is done in 64-bit here; i32 wrap-around (release Rust) lands on a
 * negative index, i.e. the same `None`. */
static inline int o_get(const float* grid, int32_t rows, int32_t cols, int64_t x, int64_t y,
                        float* out)
{
    if (x < 0 || y < 0 || y >= rows || x >= cols) return 0;
    *out = grid[(size_t)y * (size_t)cols + (size_t)x];
    return 1;
}

/* util.rs:44-58 */
float oracle_bilinear(const float* grid, int32_t rows, int32_t cols, float px, float py)
{
    const float FMAX = 1e12f; /* util.rs:45 */
    float bx = floorf(px), by = floorf(py);   /* :47 */
    float tx = px - bx, ty = py - by;         /* :48 */
    float sx = 1.0f - tx, sy = 1.0f - ty;     /* :49 */
    int64_t ix = o_f32_as_i32(bx), iy = o_f32_as_i32(by); /* :50 */

    float g, y = 0.0f;                        /* :52 */
    g = FMAX; o_get(grid, rows, cols, ix, iy, &g);         y += sy * sx * g; /* :53 */
    g = FMAX; o_get(grid, rows, cols, ix + 1, iy, &g);     y += sy * tx * g; /* :54 */
    g = FMAX; o_get(grid, rows, cols, ix, iy + 1, &g);     y += ty * sx * g; /* :55 */
    g = FMAX; o_get(grid, rows, cols, ix + 1, iy + 1, &g); y += ty * tx * g; /* :56 */
    return y;
}

/* util.rs:61-75.  First digit of uRC = row (y) offset, second = column (x). */
void oracle_sobel_filter(const float* grid, int32_t rows, int32_t cols, float px, float py,
                         float* out_xy)
{
    float u00 = oracle_bilinear(grid, rows, cols, px + -1.0f, py + -1.0f);
    float u01 = oracle_bilinear(grid, rows, cols, px + 0.0f, py + -1.0f);
    float u02 = oracle_bilinear(grid, rows, cols, px + 1.0f, py + -1.0f);
    float u10 = oracle_bilinear(grid, rows, cols, px + -1.0f, py + 0.0f);
    float u12 = oracle_bilinear(grid, rows, cols, px + 1.0f, py + 0.0f);
    float u20 = oracle_bilinear(grid, rows, cols, px + -1.0f, py + 1.0f);
    float u21 = oracle_bilinear(grid, rows, cols, px + 0.0f, py + 1.0f);
    float u22 = oracle_bilinear(grid, rows, cols, px + 1.0f, py + 1.0f);
    out_xy[0] = u00 + u10 + u10 + u20 - u02 - u12 - u12 - u22; /* :72 */
    out_xy[1] = u00 + u01 + u01 + u02 - u20 - u21 - u21 - u22; /* :73 */
}

/* test hook: sobel_filter + bilinear (util.rs:44-75) at n grid-coordinate points */
void oracle_sample_many(const float* grid, int32_t rows, int32_t cols, const float* px,
                        const float* py, float* grad_xy, float* centre, uint32_t n)
{
    for (uint32_t k = 0; k < n; ++k) {
        oracle_sobel_filter(grid, rows, cols, px[k], py[k], grad_xy + 2 * k);
        centre[k] = oracle_bilinear(grid, rows, cols, px[k], py[k]);
    }
}

/* util.rs:92-103.  The degenerate branch really is `a - line[0]` upstream. */
void oracle_distance_from_line(float px, float py, const float* l, float* out_xy)
{
    ovec2 l0 = ov(l[0], l[1]), l1 = ov(l[2], l[3]);
    ovec2 a = ov_sub(ov(px, py), l0);
    ovec2 b = ov_sub(l1, l0);
    float b_len2 = ov_length_squared(b);
    ovec2 r;
    if (b_len2 == 0.0f) {
        r = ov_sub(a, l0);
    } else {
        /* Rust f32::max/min = IEEE minNum/maxNum (NaN loses), same as fmaxf/fminf */
        float t = fminf(fmaxf(ov_dot(a, b) / b_len2, 0.0f), 1.0f);
        r = ov_sub(a, ov_scale(b, t)); /* `t * b` lane-wise */
    }
    out_xy[0] = r.x;
    out_xy[1] = r.y;
}

/* util.rs:106-111 */
void oracle_line_with_width(const float* l, float width, float* out)
{
    ovec2 l0 = ov(l[0], l[1]), l1 = ov(l[2], l[3]);
    ovec2 a = ov_normalize(ov_sub(l1, l0));
    ovec2 b = ov_scale(ov_scale(ov(a.y, -a.x), 0.5f), width);
    ovec2 v0 = ov_sub(l0, b), v1 = ov_add(l0, b), v2 = ov_add(l1, b), v3 = ov_sub(l1, b);
    out[0] = v0.x; out[1] = v0.y; out[2] = v1.x; out[3] = v1.y;
    out[4] = v2.x; out[5] = v2.y; out[6] = v3.x; out[7] = v3.y;
}

/* ---- build-owned RNG --------------------------------------------------------
 * The reference uses the unseeded global `fastrand` generator, so no
 * reference stream exists to reproduce.  This is the build's documented generator:
 * WyRand step, 24-bit mantissa f32, 53-bit f64, Irwin-Hall(12) normal approximation.
 * The product host (pedoni_amd/csrc/host) implements the same specification.
 */
uint64_t oracle_rng_next(uint64_t* s)
{
    *s += 0xa0761d6478bd642fULL;
    __uint128_t t = (__uint128_t)(*s) * (__uint128_t)(*s ^ 0xe7037ed1a0b428dbULL);
    return (uint64_t)(t >> 64) ^ (uint64_t)t;
}
float oracle_rng_f32(uint64_t* s) { return (float)(oracle_rng_next(s) >> 40) * 0x1.0p-24f; }
double oracle_rng_f64(uint64_t* s) { return (double)(oracle_rng_next(s) >> 11) * 0x1.0p-53; }
float oracle_rng_normal_approx(uint64_t* s, float mu, float sigma)
{
    float acc = 0.0f;
    for (int i = 0; i < 12; ++i) acc += oracle_rng_f32(s);
    return mu + sigma * (acc - 6.0f);
}

/* util.rs:78-89 (Knuth product of uniforms, f64) */
int32_t oracle_poisson(uint64_t* s, double lambda)
{
    int32_t y = 0;
    double x = oracle_rng_f64(s);
    double exp_lambda = exp(-lambda);
    while (x >= exp_lambda) {
        x *= oracle_rng_f64(s);
        y += 1;
    }
    return y;
}

/* ---- expf restated -----------------------------------------------------------
 * Rust's f32::exp lowers to the platform libm's expf (glibc on linux-gnu).  glibc
 * >= 2.27 evaluates expf in double with a 32-entry 2^(i/32) table and a cubic
 * (published algorithm: ARM optimized-routines / glibc sysdeps/ieee754/flt-32/e_expf.c,
 * x86-64 multiarch FMA variant).  The HIP kernels evaluate exactly this sequence in
 * f64 so that device results are bit-identical to the host libm; tests/ compares this
 * restatement with libm expf bit-for-bit.  The oracle's model path itself calls libm
 * expf, as the reference does.
 */
static const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

float oracle_expf_restated(float x)
{
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N;
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    const double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    const double C2 = 0x1.62e42ff0c52d6p-1 / N;

    uint32_t ux;
    memcpy(&ux, &x, 4);
    uint32_t abstop = (ux >> 20) & 0x7ff;
    if (abstop >= 0x42b) { /* |x| >= 88 or NaN */
        if (ux == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    double r = z - kd;
    uint64_t t = EXP2F_TAB[ki % 32];
    t += ki << (52 - 5);
    double s;
    memcpy(&s, &t, 8);
    double zz = fma(C0, r, C1);
    double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    y = y * s;
    return (float)y;
}
