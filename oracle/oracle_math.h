/* oracle_math.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar restatement of the glam 0.29.2 `Vec2` arithmetic that the reference's
 * hot path leans on (third-party crate, pinned in /root/reference/Cargo.lock:525-528,
 * source NOT under /root/reference).  Each helper spells out the operation order
 * glam documents so that gcc (-O2 -ffp-contract=off -fno-fast-math, SSE2 scalar
 * float) evaluates exactly what rustc/LLVM would: Rust never contracts a*b+c.
 *
 * parity unpinned: glam semantics are restated from its published documentation;
 * the reference's own tests pin only `length()` indirectly (util.rs:152-153).
 */
#ifndef PEDONI_ORACLE_MATH_H
#define PEDONI_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>

typedef struct { float x, y; } ovec2;

static inline ovec2 ov(float x, float y) { ovec2 r = { x, y }; return r; }
static inline ovec2 ov_add(ovec2 a, ovec2 b) { return ov(a.x + b.x, a.y + b.y); }
static inline ovec2 ov_sub(ovec2 a, ovec2 b) { return ov(a.x - b.x, a.y - b.y); }
static inline ovec2 ov_neg(ovec2 a) { return ov(-a.x, -a.y); }
/* Vec2 * f32 and f32 * Vec2 are both lane-wise products. */
static inline ovec2 ov_scale(ovec2 a, float s) { return ov(a.x * s, a.y * s); }
/* Vec2 / f32 is a lane-wise IEEE division (NOT a multiply by the reciprocal). */
static inline ovec2 ov_div(ovec2 a, float s) { return ov(a.x / s, a.y / s); }
/* glam: dot = (x * rhs.x) + (y * rhs.y) */
static inline float ov_dot(ovec2 a, ovec2 b) { return (a.x * b.x) + (a.y * b.y); }
static inline float ov_length_squared(ovec2 a) { return ov_dot(a, a); }
static inline float ov_length(ovec2 a) { return sqrtf(ov_dot(a, a)); }
/* glam: normalize = self * self.length().recip(); no zero check in release. */
static inline ovec2 ov_normalize(ovec2 a) { return ov_scale(a, 1.0f / ov_length(a)); }
/* glam: normalize_or_zero -> try_normalize: rcp finite && rcp > 0 ? self*rcp : ZERO */
static inline ovec2 ov_normalize_or_zero(ovec2 a)
{
    float rcp = 1.0f / ov_length(a);
    if (isfinite(rcp) && rcp > 0.0f) return ov_scale(a, rcp);
    return ov(0.0f, 0.0f);
}
/* glam: clamp_length_max: if len_sq > max*max { max * (self / sqrt(len_sq)) } else self */
static inline ovec2 ov_clamp_length_max(ovec2 a, float max)
{
    float length_sq = ov_length_squared(a);
    if (length_sq > max * max) {
        ovec2 q = ov_div(a, sqrtf(length_sq));
        return ov(max * q.x, max * q.y);
    }
    return a;
}
/* glam 0.29: lerp = self * (1.0 - s) + rhs * s   (spawn placement only, lib.rs:43,76) */
static inline ovec2 ov_lerp(ovec2 a, ovec2 b, float s)
{
    return ov_add(ov_scale(a, 1.0f - s), ov_scale(b, s));
}

/* Rust `f32 as i32`: truncate toward zero, saturate, NaN -> 0. */
static inline int32_t o_f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
/* Rust `f32 as usize` (64-bit): truncate, saturate, negative/NaN -> 0. */
static inline uint64_t o_f32_as_usize(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)v;
}

#endif
