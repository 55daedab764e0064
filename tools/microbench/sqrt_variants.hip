// sqrt_variants.hip -- exhaustive comparison of shorter correctly-rounded-sqrt candidates against
// device_math.hpp's sqrt_core (itself checked against the host's sqrtf by tools/exhaustive_sqrt.py)
// over every float of sqrt_core's domain [2^-96, +inf).
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o /tmp/sqrt_variants tools/microbench/sqrt_variants.hip && /tmp/sqrt_variants
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float sqrt_core(float a)
{
    float s = __builtin_amdgcn_sqrtf(a);
    float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    float r_dn = __builtin_fmaf(-s_dn, s, a);
    float r_up = __builtin_fmaf(-s_up, s, a);
    s = r_dn <= 0.0f ? s_dn : s;
    s = r_up > 0.0f ? s_up : s;
    return s;
}
// A: rsq, one Markstein step
__device__ __forceinline__ float cand_a(float a)
{
    float y = __builtin_amdgcn_rsqf(a);
    float g = a * y, h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, a);
    return __builtin_fmaf(d, h, g);
}
// B: v_sqrt, residual, correction with h = 0.5 * rsq
__device__ __forceinline__ float cand_b(float a)
{
    float g = __builtin_amdgcn_sqrtf(a);
    float h = 0.5f * __builtin_amdgcn_rsqf(a);
    float d = __builtin_fmaf(-g, g, a);
    return __builtin_fmaf(d, h, g);
}
// C: rsq with a refined h (Goldschmidt step), then the Markstein step
__device__ __forceinline__ float cand_c(float a)
{
    float y = __builtin_amdgcn_rsqf(a);
    float g = a * y, h = 0.5f * y;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, a);
    return __builtin_fmaf(d, h, g);
}

// reciprocal: rcp + one Newton step (device_math.hpp recip_refined) against the full IEEE 1 / d
__global__ void sweep_rcp(uint32_t lo, uint32_t hi, unsigned long long* bad)
{
    unsigned long long b0 = 0, b1 = 0;
    for (uint64_t u = (uint64_t)lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u < hi;
         u += (uint64_t)gridDim.x * blockDim.x) {
        float d = __uint_as_float((uint32_t)u);
        float y = __builtin_amdgcn_rcpf(d);
        float e = __builtin_fmaf(-d, y, 1.0f);
        float y1 = __builtin_fmaf(e, y, y);
        uint32_t want = __float_as_uint(1.0f / d);       // hipcc's correctly rounded division
        b0 += __float_as_uint(y1) != want;
        b1 += __float_as_uint(y) != want;
    }
    if (b0) atomicAdd(&bad[4], b0);
    if (b1) atomicAdd(&bad[5], b1);
}

// 1 / sqrt_core(a), correctly rounded, WITHOUT a second transcendental: Newton step on the rsq the
// square root started from (against rcp + Newton on s, itself == 1 / s by sweep_rcp)
__global__ void sweep_rsq_rcp(uint32_t lo, uint32_t hi, unsigned long long* bad)
{
    unsigned long long b0 = 0;
    for (uint64_t u = (uint64_t)lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u < hi;
         u += (uint64_t)gridDim.x * blockDim.x) {
        float a = __uint_as_float((uint32_t)u);
        float y0 = __builtin_amdgcn_rsqf(a);
        float g = a * y0, h = 0.5f * y0;
        float s = __builtin_fmaf(__builtin_fmaf(-g, g, a), h, g);     // cand_a
        float y = __builtin_fmaf(__builtin_fmaf(-s, y0, 1.0f), y0, y0);
        b0 += __float_as_uint(y) != __float_as_uint(1.0f / s);
    }
    if (b0) atomicAdd(&bad[6], b0);
}

__global__ void sweep(uint32_t lo, uint32_t hi, unsigned long long* bad)
{
    unsigned long long ba = 0, bb = 0, bc = 0, bd = 0;
    for (uint64_t u = (uint64_t)lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u < hi;
         u += (uint64_t)gridDim.x * blockDim.x) {
        float a = __uint_as_float((uint32_t)u);
        uint32_t want = __float_as_uint(sqrt_core(a));
        ba += __float_as_uint(cand_a(a)) != want;
        bb += __float_as_uint(cand_b(a)) != want;
        bc += __float_as_uint(cand_c(a)) != want;
        bd += __float_as_uint(__builtin_amdgcn_sqrtf(a)) != want;   // control: the bare 1-ulp instruction
    }
    if (ba) atomicAdd(&bad[0], ba);
    if (bb) atomicAdd(&bad[1], bb);
    if (bc) atomicAdd(&bad[2], bc);
    if (bd) atomicAdd(&bad[3], bd);
}

int main()
{
    unsigned long long* d_bad;
    hipMalloc((void**)&d_bad, 8 * sizeof *d_bad);
    hipMemset(d_bad, 0, 8 * sizeof *d_bad);
    const uint32_t lo = 0x0F800000u, hi = 0x7F800000u;   // [2^-96, +inf)
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, lo, hi, d_bad);
    const uint32_t rlo = 0x01000000u, rhi = 0x7E800000u;  // [2^-125, 2^126): d and 1 / d both normal
    hipLaunchKernelGGL(sweep_rsq_rcp, dim3(4096), dim3(256), 0, 0, lo, 0x7E800000u, d_bad);
    hipLaunchKernelGGL(sweep_rcp, dim3(4096), dim3(256), 0, 0, rlo, rhi, d_bad);
    unsigned long long bad[8];
    hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
    printf("floats checked: %u\n", hi - lo);
    printf("A (rsq + one Markstein step, 5 instructions): %llu mismatches\n", bad[0]);
    printf("B (sqrt + rsq + residual step, 5 instructions, 2 transcendental): %llu mismatches\n", bad[1]);
    printf("C (rsq + Goldschmidt + Markstein, 8 instructions): %llu mismatches\n", bad[2]);
    printf("control (bare v_sqrt_f32): %llu mismatches\n", bad[3]);
    printf("reciprocal, %u floats: rcp + one Newton step %llu mismatches against 1 / d (bare v_rcp_f32: %llu)\n", rhi - rlo, bad[4], bad[5]);
    printf("1 / sqrt(a) from the rsq + one Newton step: %llu mismatches against 1 / s\n", bad[6]);
    return 0;
}
