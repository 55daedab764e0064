// device_math.hpp -- gfx950 device primitives of the social-force path.
//
// Every function spells out the fp32 operation order of the reference CPU path
// (glam Vec2 arithmetic as used by pedoni-simulator/src/models/sfm.rs, util.rs,
// field.rs).  The translation unit is built with -ffp-contract=off so a*b+c is never
// fused, f32 division and sqrt are hipcc's correctly rounded forms, and f32 denormals
// are kept: in PEDONI_MATH_EXACT every value below is bit-identical to what rustc
// produces on x86-64.  MODE = 1 (PEDONI_MATH_FAST) swaps division / sqrt / exp for the
// 1-ulp hardware approximations (v_rcp_f32, v_rsq_f32, v_sqrt_f32, v_exp_f32).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pedoni {

struct v2 { float x, y; };

__device__ __forceinline__ v2 mk(float x, float y) { v2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ v2 operator+(v2 a, v2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ v2 operator-(v2 a, v2 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ v2 operator-(v2 a) { return mk(-a.x, -a.y); }
__device__ __forceinline__ v2 operator*(v2 a, float s) { return mk(a.x * s, a.y * s); }
__device__ __forceinline__ float dot(v2 a, v2 b) { return (a.x * b.x) + (a.y * b.y); }

// ---- division / sqrt / exp by math mode -------------------------------------------
template <int MODE> __device__ __forceinline__ float fdiv(float a, float b)
{
    if constexpr (MODE == 0) return a / b;             // IEEE, correctly rounded
    else return a * __builtin_amdgcn_rcpf(b);
}
template <int MODE> __device__ __forceinline__ float frcp(float b)
{
    if constexpr (MODE == 0) return 1.0f / b;
    else return __builtin_amdgcn_rcpf(b);
}
// Correctly rounded sqrt.  hipcc's expansion of sqrtf is v_sqrt_f32 (1 ulp) plus a one-ulp
// correction from two fma residuals, wrapped in a 2^32 rescale for x < 2^-96 and a class test
// for 0 / inf / NaN: 16 VALU instructions.  For 2^-96 <= x < inf sqrt_core below gives the very
// same bits in 5; anything else (0, denormal, tiny, negative, inf, NaN) takes the compiler's
// form.  Checked against the host's sqrtf on every non-negative float (tools/exhaustive_sqrt.py).
__device__ __noinline__ float sqrt_generic(float a) { return __builtin_sqrtf(a); }
constexpr uint32_t SQRT_CORE_LO = 0x0F800000u;   // bits of 2^-96
constexpr uint32_t SQRT_CORE_SPAN = 0x70000000u; // up to, not including, +inf
// < SQRT_CORE_SPAN iff sqrt_core(a) is valid
__device__ __forceinline__ uint32_t sqrt_core_measure(float a) { return __float_as_uint(a) - SQRT_CORE_LO; }
__device__ __forceinline__ float sqrt_core(float a)
{
    // v_rsq_f32 (1 ulp) and ONE residual step: g = a * y, h = y / 2, s = g + (a - g * g) * h.
    // Equal to the correctly rounded square root -- the bits of the compiler's own expansion
    // (v_sqrt_f32 and a test of both neighbours, 9 instructions) -- for EVERY float in
    // [2^-96, +inf): tools/microbench/sqrt_variants.hip compares all 1 879 048 192 of them.
    float y = __builtin_amdgcn_rsqf(a);
    float g = a * y, h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, a);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_rn(float a)
{
    if (__builtin_expect(sqrt_core_measure(a) < SQRT_CORE_SPAN, 1)) return sqrt_core(a);
    return sqrt_generic(a);
}
template <int MODE> __device__ __forceinline__ float fsqrt(float a)
{
    if constexpr (MODE == 0) return sqrt_rn(a);         // correctly rounded
    else return __builtin_amdgcn_sqrtf(a);
}

// x / d for the two constant divisors of the force law (0.3 and 0.2), correctly rounded:
// with zh = RN(1/d) and zl = RN(1/d - zh), fma(x, zh, x * zl) == RN(x / d) for EVERY float
// with 2^-100 <= |x| <= 2^100 (checked exhaustively on the host against IEEE division,
// 1.69e9 values per divisor); anything else (0, tiny, huge, inf, NaN) takes the division.
template <int MODE> __device__ __forceinline__ float div_03(float x)
{
    if constexpr (MODE != 0) return x * __builtin_amdgcn_rcpf(0.3f);
    uint32_t u = __float_as_uint(x) & 0x7fffffffu;
    if (u - 0x0D800000u <= 0x647FFFFFu) return __builtin_fmaf(x, 0x1.aaaaaap+1f, x * -0x1.c71c6ep-25f);
    return x / 0.3f;
}
template <int MODE> __device__ __forceinline__ float div_02(float x)
{
    if constexpr (MODE != 0) return x * __builtin_amdgcn_rcpf(0.2f);
    uint32_t u = __float_as_uint(x) & 0x7fffffffu;
    if (u - 0x0D800000u <= 0x647FFFFFu) return __builtin_fmaf(x, 0x1.4p+2f, x * -0x1.4p-24f);
    return x / 0.2f;
}

// 2^(i/32) table of glibc's expf (bits minus i<<47), sysdeps/ieee754/flt-32/e_exp2f_data.c
__device__ const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

// the arithmetic of glibc's expf for |x| < 88 (no special cases)
__device__ __forceinline__ float exp_glibc_core(float x, const uint64_t* tab)
{
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N;
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    const double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    const double C2 = 0x1.62e42ff0c52d6p-1 / N;
    double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= SHIFT;
    double r = z - kd;
    uint64_t t = tab[ki & 31];
    t += ki << 47;
    double s = __longlong_as_double((long long)t);
    double zz = __builtin_fma(C0, r, C1);
    double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(zz, r2, y);
    y = y * s;
    return (float)y;
}


// Rust's f32::exp is the host libm's expf.  glibc >= 2.27 computes it in double with
// the table above and a cubic, using FMA on x86-64 (multiarch variant); this replays
// that sequence operation for operation in f64, so the result equals the host's bit
// for bit (verified against glibc 2.35 on all 2.2e9 floats in [-104, 88] but two).
__device__ __forceinline__ float exp_glibc(float x, const uint64_t* tab)
{
    uint32_t ux = __float_as_uint(x);
    uint32_t abstop = (ux >> 20) & 0x7ff;
    if (abstop >= 0x42b) { // |x| >= 88 or NaN
        if (ux == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_inff();
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    return exp_glibc_core(x, tab);
}

template <int MODE> __device__ __forceinline__ float fexp(float x, const uint64_t* tab)
{
    if constexpr (MODE == 0) return exp_glibc(x, tab);
    else return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}

// ---- glam helpers -------------------------------------------------------------------
template <int MODE> __device__ __forceinline__ float length(v2 a) { return fsqrt<MODE>(dot(a, a)); }
// glam normalize: self * length().recip()
template <int MODE> __device__ __forceinline__ v2 normalize(v2 a)
{
    if constexpr (MODE == 0) return a * (1.0f / length<0>(a));
    else return a * __builtin_amdgcn_rsqf(dot(a, a));
}
template <int MODE> __device__ __forceinline__ v2 vdiv(v2 a, float s)
{
    if constexpr (MODE == 0) return mk(a.x / s, a.y / s);
    else { float r = __builtin_amdgcn_rcpf(s); return mk(a.x * r, a.y * r); }
}
// glam normalize_or_zero (sfm.rs:199)
__device__ __forceinline__ v2 normalize_or_zero(v2 a)
{
    float rcp = 1.0f / length<0>(a);
    if (__builtin_isfinite(rcp) && rcp > 0.0f) return a * rcp;
    return mk(0.0f, 0.0f);
}

// Rust `f32 as i32`: truncate toward zero, saturate, NaN -> 0 (the hardware's V_CVT_I32_F32
// has exactly these semantics, and using it directly was tried: it removes the branches, the
// scheduler then overlaps many more texel loads, VGPRs rise 82 -> 110 and the force kernel
// loses a wave per SIMD and ~4 % -- so the explicit form stays).
__device__ __forceinline__ int32_t f32_as_i32(float v)
{
#ifdef PEDONI_CVT_HW   // A/B switch (tools/ab_flags.sh "-DPEDONI_CVT_HW"): the bare instruction, see above
    int32_t r;
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
#endif
}

// ---- field sampling (util.rs:44-75, field.rs:235-258) -----------------------------
struct FieldView {
    const float* distance_map;
    // The potential maps live in ONE allocation, map m at pot_base + m * pot_stride: a map's address is
    // arithmetic on the agent's destination, not a pointer fetched from memory -- one dependent round trip
    // less at the head of every force-kernel wave (and the address is a global one to the compiler as it is).
    const float* pot_base;
    int64_t pot_stride;                 // floats from one map to the next
    int32_t rows, cols;
    float unit;
    uint32_t n_maps;
    // unit is a power of two (the default 0.25 is): x / unit == x * inv_unit for every x --
    // scaling by a power of two is exact, and overflows / rounds into the denormals exactly
    // where the division does
    float inv_unit;
    int32_t unit_pow2;
    // A band of a sharded run may hold only the texel rows [y_lo, y_hi) of every map (the map
    // pointers are then biased by -y_lo rows, so indexing stays in full-field coordinates).
    // Whole field: y_lo = 0, y_hi = rows.  A texel inside the field but outside the slice is
    // never read: it returns the out-of-field value and raises `status` bit 2 (loudly wrong
    // instead of a fault) -- the slice is sized so that no agent of the band gets there.
    int32_t y_lo, y_hi;
    uint32_t* status;
};

__device__ __forceinline__ const float* potential_map(const FieldView& f, uint32_t waypoint)
{
    return f.pot_base + (int64_t)waypoint * f.pot_stride;
}

// shape of one map as the sampling functions see it
struct MapDims {
    int32_t rows, cols, y_lo, y_hi;
    uint32_t* status;
};
__device__ __forceinline__ MapDims dims_of(const FieldView& f)
{
    MapDims d;
    d.rows = f.rows; d.cols = f.cols; d.y_lo = f.y_lo; d.y_hi = f.y_hi; d.status = f.status;
    return d;
}
constexpr uint32_t STATUS_FIELD_SLICE = 4u; // a texel outside the uploaded rows of the maps was asked for

// A map pointer that is not visibly derived from a kernel argument is a FLAT pointer to the compiler: its
// loads would be flat_load_dword -- never merged into wider loads, and counted on the LDS counter
// as well as the vector-memory one.  Every map lives in global memory: say so.
typedef const __attribute__((address_space(1))) float* MapPtr;
__device__ __forceinline__ MapPtr as_map(const float* g) { return (MapPtr)g; }

// util.rs:30-36 + :53-56: texel or 1e12 when the index is negative / out of shape
__device__ __forceinline__ float texel(const float* g_, const MapDims& m, int64_t x, int64_t y)
{
    MapPtr g = as_map(g_);
    if (x < 0 || y < 0 || y >= m.rows || x >= m.cols) return 1e12f;
    if (y < m.y_lo || y >= m.y_hi) {                     // in the field, not in this band's slice
        atomicOr(m.status, STATUS_FIELD_SLICE);
        return 1e12f;
    }
    return g[(int64_t)y * (int64_t)m.cols + x];
}

// util.rs:44-58
__device__ __forceinline__ float bilinear(const float* g_, const MapDims& m, float px, float py)
{
    MapPtr g = as_map(g_);
    const int32_t cols = m.cols;
    float bx = __builtin_floorf(px), by = __builtin_floorf(py);
    float tx = px - bx, ty = py - by;
    float sx = 1.0f - tx, sy = 1.0f - ty;
    int64_t ix = f32_as_i32(bx), iy = f32_as_i32(by);
    float g00, g01, g10, g11;
    if (ix >= 0 && iy >= m.y_lo && ix + 1 < cols && iy + 1 < m.y_hi) {   // all four texels in bounds
        MapPtr r0 = g + ((int64_t)iy * (int64_t)cols + ix);
        g00 = r0[0]; g01 = r0[1]; g10 = r0[cols]; g11 = r0[cols + 1];
    } else {
        g00 = texel(g_, m, ix, iy);     g01 = texel(g_, m, ix + 1, iy);
        g10 = texel(g_, m, ix, iy + 1); g11 = texel(g_, m, ix + 1, iy + 1);
    }
    float y = 0.0f;
    y += sy * sx * g00;
    y += sy * tx * g01;
    y += ty * sx * g10;
    y += ty * tx * g11;
    return y;
}

// util.rs:61-75 (first digit = row offset, second = column offset)
__device__ __forceinline__ v2 sobel(const float* g, const MapDims& m, float px, float py)
{
    float u00 = bilinear(g, m, px + -1.0f, py + -1.0f);
    float u01 = bilinear(g, m, px + 0.0f, py + -1.0f);
    float u02 = bilinear(g, m, px + 1.0f, py + -1.0f);
    float u10 = bilinear(g, m, px + -1.0f, py + 0.0f);
    float u12 = bilinear(g, m, px + 1.0f, py + 0.0f);
    float u20 = bilinear(g, m, px + -1.0f, py + 1.0f);
    float u21 = bilinear(g, m, px + 0.0f, py + 1.0f);
    float u22 = bilinear(g, m, px + 1.0f, py + 1.0f);
    return mk(u00 + u10 + u10 + u20 - u02 - u12 - u12 - u22,
              u00 + u01 + u01 + u02 - u20 - u21 - u21 - u22);
}

// ---- shared-patch sampling ---------------------------------------------------------------
// util::sobel_filter takes 8 bilinear samples at p + (+-1 | 0, +-1 | 0); with the centre
// sample of get_obstacle_distance that is a 3 x 3 stencil of bilinear taps whose texels
// all lie in one 4 x 4 patch.  The reference recomputes floor / fraction / 4 gathers per
// tap; here the three per-axis (floor, fraction) sets are computed once -- with the very
// same fp32 operations, so every tap value is bit-identical -- and the 16 texels are
// loaded once.  If rounding of p +- 1 moves a tap's floor off the patch (or p is NaN /
// huge), the literal per-tap path below is used instead.
struct AxisTaps {
    float s[3], t[3]; // weights of the tap at offset -1, 0, +1
    int32_t i[3];     // its base texel index
};

__device__ __forceinline__ AxisTaps axis_taps(float p)
{
    AxisTaps a;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float q = p + (float)(k - 1);          // pos + vec2(-1 | 0 | 1, ..)   util.rs:62-69
        float b = __builtin_floorf(q);         // util.rs:47
        a.t[k] = q - b;                        // :48
        a.s[k] = 1.0f - a.t[k];                // :49
        a.i[k] = f32_as_i32(b);                // :50
    }
    return a;
}

// u[r][c] = bilinear(g, p + (c - 1, r - 1)); returns false when the patch form does not apply
__device__ __forceinline__ bool stencil_taps(const float* __restrict__ g_, const MapDims& m, float px,
                                             float py, float (&u)[3][3])
{
    MapPtr g = as_map(g_);
    const int32_t cols = m.cols;
    AxisTaps ax = axis_taps(px), ay = axis_taps(py);
    int64_t x0 = ax.i[0], y0 = ay.i[0];
    if (ax.i[1] != x0 + 1 || ax.i[2] != x0 + 2 || ay.i[1] != y0 + 1 || ay.i[2] != y0 + 2)
        return false;
    float P[4][4];
    if (x0 >= 0 && y0 >= m.y_lo && x0 + 3 < cols && y0 + 3 < m.y_hi) {
        MapPtr row = g + (y0 * (int64_t)cols + x0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < 4; ++c) P[r][c] = row[c];
            row += cols;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) P[r][c] = texel(g_, m, x0 + c, y0 + r);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float y = 0.0f;                              // util.rs:52-56
            y += ay.s[r] * ax.s[c] * P[r][c];
            y += ay.s[r] * ax.t[c] * P[r][c + 1];
            y += ay.t[r] * ax.s[c] * P[r + 1][c];
            y += ay.t[r] * ax.t[c] * P[r + 1][c + 1];
            u[r][c] = y;
        }
    return true;
}

// sobel_filter (util.rs:61-75) and, optionally, the centre sample (field.rs:242-245)
__device__ __forceinline__ v2 sobel_fast(const float* __restrict__ g, const MapDims& m, float px, float py,
                                         float* centre)
{
    float u[3][3];
    if (!stencil_taps(g, m, px, py, u)) {
        if (centre) *centre = bilinear(g, m, px, py);
        return sobel(g, m, px, py);
    }
    if (centre) *centre = u[1][1];
    return mk(u[0][0] + u[1][0] + u[1][0] + u[2][0] - u[0][2] - u[1][2] - u[1][2] - u[2][2],
              u[0][0] + u[0][1] + u[0][1] + u[0][2] - u[2][0] - u[2][1] - u[2][1] - u[2][2]);
}

// field.rs:236,243,250,256: position / unit - 0.5 (the division is IEEE in both modes:
// it decides which texels are read)
__device__ __forceinline__ v2 field_coord(const FieldView& f, v2 pos)
{
    if (f.unit_pow2) return mk(pos.x * f.inv_unit - 0.5f, pos.y * f.inv_unit - 0.5f);
    return mk(pos.x / f.unit - 0.5f, pos.y / f.unit - 0.5f);
}

// ---- hot-path division ------------------------------------------------------------------
// hipcc expands a / b to v_div_scale x2, v_rcp, a Newton step on the reciprocal, two
// residual corrections of the quotient, v_div_fmas, v_div_fixup.  Where neither operand
// needs rescaling (b normal, |a| >= 2^-103, |a / b| in [2^-126, 2^96)) and no operand is
// 0 / inf / NaN, scale, fmas and fixup are identities and what remains is the arithmetic
// below -- the same instructions on the same values, hence the same bits -- and the
// reciprocal part is shared by quotients with one denominator.  pair_force_hot guarantees
// that domain through its folded range test (it rejects quotients below 2^-50, which covers
// every numerator small enough to be rescaled).
// Two things about that sequence on gfx950, both established by exhaustive runs on the device
// (tools/microbench/sqrt_variants.hip, div_variants.hip) and used below:
//  - the refined reciprocal y = fma(fma(-d, rcp(d), 1), rcp(d), rcp(d)) IS RN(1 / d) for every d in
//    [2^-125, 2^126) (all 2 105 540 608 floats), so 1 / d needs no quotient correction at all;
//  - with that y, ONE quotient correction already gives the final quotient: the second never
//    changes it, for all 2^23 x 2^23 pairs of significands (in the domain above no operand or
//    residual under- or overflows, every operation scales exactly with powers of two, and the
//    quotient's significand depends on the two significands only).
struct Recip { float d, y; };
__device__ __forceinline__ Recip recip_refined(float d)
{
    float y = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, y, 1.0f);
    Recip r;
    r.d = d;
    r.y = __builtin_fmaf(e, y, y);
    return r;
}
__device__ __forceinline__ float div_core(float n, Recip r)
{
    float q = n * r.y;
    float e = __builtin_fmaf(-r.d, q, n);
    return __builtin_fmaf(e, r.y, q);
}

// ---- pair force (sfm.rs:131-153) -----------------------------------------------------
#define PEDONI_COS_PHI (-0.17364817766693036f) /* sfm.rs:16 */

// |v_i| * 0.1 of sfm.rs:144 depends on the NEIGHBOUR only, and every agent is a neighbour in
// ~13 pairs per tick: it is computed once per agent, by the sort pass that moves the agent
// (place / reorder kernels), and stored beside its velocity ({vx, vy, vl, desired_speed}).
//   MODE 0: sqrt_core(|v|^2) * 0.1 -- the very operations pair_force_hot ran per pair -- when
//           |v|^2 lies in the hot form's range [2^-96, 2^40); otherwise -1: a negative vl
//           sends the pair to the generic path, which recomputes from the velocity itself
//   MODE 1: v_sqrt_f32(|v|^2) * 0.1, what the fast pair force computed inline
template <int MODE> __device__ __forceinline__ float neighbour_vl(v2 vel_i)
{
    const float v_sq = dot(vel_i, vel_i);
    if constexpr (MODE == 0) {
        if (sqrt_core_measure(v_sq) < 0x44000000u) return sqrt_core(v_sq) * 0.1f;
        return -1.0f;
    } else {
        return __builtin_amdgcn_sqrtf(v_sq) * 0.1f;
    }
}

// force on an agent from a neighbour, given difference = pos - pos_i with
// |difference|^2 <= 4 already established (sfm.rs:137-153), before the field-of-view test
template <int MODE>
__device__ __forceinline__ v2 pair_force_raw(v2 difference, v2 vel_i, float vl_pre, const uint64_t* tab,
                                             bool& ill_conditioned)
{
    float distance_squared = dot(difference, difference); // :132
    float distance = fsqrt<MODE>(distance_squared);      // :137
    v2 direction = difference * frcp<MODE>(distance);    // :138 normalize()

    v2 t1 = difference - vel_i * 0.1f;                   // :141
    float t1_length = length<MODE>(t1);                  // :142
    float t2 = distance + t1_length;                     // :143
    float vl;
    if constexpr (MODE == 0) vl = length<0>(vel_i) * 0.1f;   // generic exact path: from the velocity itself
    else vl = vl_pre;                                        // neighbour_vl<1>: the same v_sqrt_f32 * 0.1
    float t2_sq = t2 * t2, b_arg = t2_sq - vl * vl;
    // a neighbour about to step onto the agent: t2 -> |v| dt and the difference cancels
    ill_conditioned = !(b_arg * 8.0f > t2_sq);
    float b = fsqrt<MODE>(b_arg) * 0.5f;                 // :144

    v2 nabla_b = vdiv<MODE>((direction + vdiv<MODE>(t1, t1_length)) * t2, 4.0f * b); // :146
    float k = (2.1f / 0.3f) * fexp<MODE>(div_03<MODE>(-b), tab);                   // :147
    return nabla_b * k;
}

// Hot form of the exact pair force: the same operations as pair_force_raw<0> and the two
// sides of the field-of-view test, with the range tests of the five square roots, of
// x / 0.3, of exp and of the five divisions folded into ONE unsigned maximum (`worst`)
// instead of a branch each: every sqrt argument must lie in [2^-96, inf), |v| in
// [2^-48, 2^20), b below 26 (so |b / 0.3| < 88) and no quotient below 2^-50; then sqrt_core,
// div_core, the fma form of x / 0.3 and exp_glibc_core ARE the generic functions.  A pair
// with anything else (zero velocity, coincident agents, NaN, ...) returns
// worst >= SQRT_CORE_SPAN and is evaluated by the generic path instead.
__device__ __forceinline__ uint32_t pair_force_hot(v2 difference, v2 e, v2 vel_i, float vl, const uint64_t* tab,
                                                   v2& force, float& lhs, float& rhs)
{
    float distance_squared = dot(difference, difference); // :132
    uint32_t worst = sqrt_core_measure(distance_squared);
    float distance = sqrt_core(distance_squared);        // :137
    v2 direction = difference * recip_refined(distance).y; // :138 normalize(): difference * RN(1 / distance)

    v2 t1 = difference - vel_i * 0.1f;                   // :141
    float t1_sq = dot(t1, t1);
    worst = max(worst, sqrt_core_measure(t1_sq));
    float t1_length = sqrt_core(t1_sq);                  // :142
    float t2 = distance + t1_length;                     // :143
    // vl = neighbour_vl<0>(vel_i): sqrt_core(|v|^2) * 0.1 with |v| in [2^-48, 2^20) -- which keeps
    // every numerator and denominator below within 2^+-96 of each other -- or -1 (sign bit set:
    // a huge measure) when |v|^2 is outside that range
    worst = max(worst, __float_as_uint(vl));
    float b_arg = t2 * t2 - vl * vl;
    worst = max(worst, sqrt_core_measure(b_arg));
    float b = sqrt_core(b_arg) * 0.5f;                   // :144
    // b >= 2^-49 by the test on b_arg; b < 26 keeps x = -b / 0.3 inside (-88, -2^-100]
    worst = max(worst, __float_as_uint(b) + (SQRT_CORE_SPAN - 0x41D00000u));

    Recip rl = recip_refined(t1_length);
    v2 q = mk(div_core(t1.x, rl), div_core(t1.y, rl));
    v2 num = (direction + q) * t2;
    Recip rb = recip_refined(4.0f * b);
    v2 nabla_b = mk(div_core(num.x, rb), div_core(num.y, rb)); // :146
    // the four quotients must be >= 2^-50 in magnitude (0x26800000): none is zero and no
    // numerator was small enough for the generic division to rescale it
    float q_min = __builtin_fminf(__builtin_fminf(__builtin_fabsf(q.x), __builtin_fabsf(q.y)),
                                  __builtin_fminf(__builtin_fabsf(nabla_b.x), __builtin_fabsf(nabla_b.y)));
    worst = max(worst, __float_as_uint(q_min) - 0x26800000u);
    float x = __builtin_fmaf(-b, 0x1.aaaaaap+1f, -b * -0x1.c71c6ep-25f); // -b / 0.3, see div_03
    float k = (2.1f / 0.3f) * exp_glibc_core(x, tab);    // :147
    force = nabla_b * k;

    lhs = dot(e, -force);                                // :149
    float f_sq = dot(force, force);
    worst = max(worst, sqrt_core_measure(f_sq));
    rhs = sqrt_core(f_sq) * PEDONI_COS_PHI;
    return worst;
}

// PEDONI_MATH_EXACT: the reference's arithmetic, bit for bit.
// PEDONI_MATH_FAST: hardware rcp / rsq / sqrt / exp (about 1e-6 relative on the force), but
// the one DISCONTINUOUS decision of the force law -- halving a force that comes from outside
// the 200-degree field of view (:149-150) -- is never left to approximate numbers: when the
// test lies within 1e-4 |f| of its boundary the pair is recomputed exactly (rare, so cheap
// even as a divergent branch).  So is a pair whose `t2^2 - (|v| dt)^2` cancels to less than
// an eighth of t2^2 (a neighbour within ~|v| dt, heading straight at the agent): there the
// 1-ulp error of sqrt/rcp would be amplified up to 1000-fold.  The goal direction `e` is
// exact in both modes.  Every agent then meets the 1e-5 bar; no decision flips.
template <int MODE>
__device__ __forceinline__ v2 pair_force_value(v2 difference, v2 e, v2 vel_i, float vl, const uint64_t* tab)
{
    bool redo;
    v2 force;
    float lhs, rhs;
    if constexpr (MODE == 0) {
        if (__builtin_expect(pair_force_hot(difference, e, vel_i, vl, tab, force, lhs, rhs) >= SQRT_CORE_SPAN, 0)) {
            force = pair_force_raw<0>(difference, vel_i, vl, tab, redo);
            lhs = dot(e, -force);
            rhs = length<0>(force) * PEDONI_COS_PHI;
        }
    } else {
        force = pair_force_raw<MODE>(difference, vel_i, vl, tab, redo);
        float len = length<MODE>(force);
        lhs = dot(e, -force);
        rhs = len * PEDONI_COS_PHI;
        // ambiguous (or NaN) field-of-view test, or a cancelling b: evaluate exactly
        if (redo || !(__builtin_fabsf(lhs - rhs) > 1e-4f * len)) {
            force = pair_force_raw<0>(difference, vel_i, vl, tab, redo);
            lhs = dot(e, -force);
            rhs = length<0>(force) * PEDONI_COS_PHI;
        }
    }
    if (lhs < rhs)                                       // :149
        force = force * 0.5f;                            // :150
    return force;
}

template <int MODE>
__device__ __forceinline__ void pair_force_from_difference(v2 difference, v2 e, v2 vel_i, float vl, v2& acc,
                                                           const uint64_t* tab)
{
    acc = acc + pair_force_value<MODE>(difference, e, vel_i, vl, tab);   // :153
}

template <int MODE>
__device__ __forceinline__ void pair_force(v2 pos, v2 e, v2 pos_i, v2 vel_i, v2& acc,
                                           const uint64_t* tab)
{
    v2 difference = pos - pos_i;                         // :131
    float distance_squared = dot(difference, difference); // :132
    if (distance_squared > 4.0f) return;                 // :133 (NaN falls through, as upstream)
    pair_force_from_difference<MODE>(difference, e, vel_i, neighbour_vl<MODE>(vel_i), acc, tab);
}

// util.rs:92-103
__device__ __forceinline__ v2 distance_from_line(v2 point, v2 l0, v2 l1)
{
    v2 a = point - l0;
    v2 b = l1 - l0;
    float b_len2 = dot(b, b);
    if (b_len2 == 0.0f) return a - l0;
    float t = __builtin_fminf(__builtin_fmaxf(dot(a, b) / b_len2, 0.0f), 1.0f);
    return a - b * t;
}

} // namespace pedoni
