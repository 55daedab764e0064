// VALU issue rate on gfx950 at 1..8 waves per SIMD, and the price of one
// phase-2 round of the force kernel (pair_force_from_difference on register operands).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I pedoni_amd/csrc -I include \
//         tools/microbench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
// Every wave records its SIMD (HW_ID / XCC_ID) and absolute s_memtime stamps; per SIMD the host
// takes (last end - first start) / (instructions issued by all its waves) -- the SIMD's issue
// rate whatever order the arbiter served its waves in (it favours the oldest, so per-wave
// medians mislead) -- and prints the median over SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "device_math.hpp"

using namespace pedoni;

constexpr int ITERS = 4000;

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ uint64_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = EXP2F_TAB[threadIdx.x];
    __syncthreads();
    int lane = threadIdx.x & 63;
    float a0 = 1.0f + lane * 1e-3f, a1 = a0 * 0.5f, a2 = a0 * 0.25f, a3 = a0 * 0.125f;
    float a4 = a0 * 1.5f, a5 = a0 * 1.25f, a6 = a0 * 1.125f, a7 = a0 * 1.0625f;
    const float b = 0.9999f, c = 1e-4f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (OP == 0) {          // 8 independent v_fma_f32
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        }
    } else if constexpr (OP == 1) {   // one dependent v_fma_f32 chain (8 per iteration)
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a0 = __builtin_fmaf(a0, b, c);
        }
    } else if constexpr (OP == 2) {   // 8 independent v_sqrt_f32
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_sqrtf(a0); a1 = __builtin_amdgcn_sqrtf(a1); a2 = __builtin_amdgcn_sqrtf(a2); a3 = __builtin_amdgcn_sqrtf(a3);
            a4 = __builtin_amdgcn_sqrtf(a4); a5 = __builtin_amdgcn_sqrtf(a5); a6 = __builtin_amdgcn_sqrtf(a6); a7 = __builtin_amdgcn_sqrtf(a7);
        }
    } else if constexpr (OP == 3) {   // 4 independent v_fma_f64 (x2 to make 8)
        for (int i = 0; i < iters; ++i) {
            d0 = __builtin_fma(d0, 0.9999, 1e-4); d1 = __builtin_fma(d1, 0.9999, 1e-4); d2 = __builtin_fma(d2, 0.9999, 1e-4); d3 = __builtin_fma(d3, 0.9999, 1e-4);
            d0 = __builtin_fma(d0, 0.9999, 1e-4); d1 = __builtin_fma(d1, 0.9999, 1e-4); d2 = __builtin_fma(d2, 0.9999, 1e-4); d3 = __builtin_fma(d3, 0.9999, 1e-4);
        }
    } else if constexpr (OP == 4) {   // one exact pair force per iteration (a phase-2 round)
        v2 acc = mk(0.0f, 0.0f);
        v2 diff = mk(0.3f + lane * 0.01f, 0.4f + lane * 0.005f), e = mk(0.8f, 0.6f), vi = mk(0.5f + lane * 0.001f, -0.3f);
        for (int i = 0; i < iters / 8; ++i) {
            pair_force_from_difference<0>(diff, e, vi, neighbour_vl<0>(vi) + 0.0f * acc.x, acc, tab);
            diff.x = __uint_as_float(__float_as_uint(diff.x) ^ ((__float_as_uint(acc.x) >> 22) & 1u));   // keep it live, keep it in range
        }
        a0 = acc.x + acc.y;
    } else if constexpr (OP == 5) {   // one fast-mode pair force per iteration
        v2 acc = mk(0.0f, 0.0f);
        v2 diff = mk(0.3f + lane * 0.01f, 0.4f + lane * 0.005f), e = mk(0.8f, 0.6f), vi = mk(0.5f + lane * 0.001f, -0.3f);
        for (int i = 0; i < iters / 8; ++i) {
            pair_force_from_difference<1>(diff, e, vi, neighbour_vl<1>(vi), acc, tab);
            diff.x = __uint_as_float(__float_as_uint(diff.x) ^ ((__float_as_uint(acc.x) >> 22) & 1u));
        }
        a0 = acc.x + acc.y;
    }
    else if constexpr (OP == 6) {   // 4 independent v_pk_fma_f32 (x2): 2 fp32 FMAs per lane each
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
        const f2 pb = {b, b}, pc = {c, c};
        for (int i = 0; i < iters; ++i) {
            p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
            p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
            p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
            p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
        }
        a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y;
    } else if constexpr (OP == 7) {   // 8 independent v_cndmask / integer adds mix: v_add_u32
        unsigned u0 = lane, u1 = lane + 1, u2 = lane + 2, u3 = lane + 3, u4 = lane + 4, u5 = lane + 5, u6 = lane + 6, u7 = lane + 7;
        for (int i = 0; i < iters; ++i) {
            u0 += u1; u1 += u2; u2 += u3; u3 += u4; u4 += u5; u5 += u6; u6 += u7; u7 += u0;
        }
        a0 = (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
    if (lane == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        unsigned long long* r = cyc + 3 * (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6));
        r[0] = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xfff0u);   // xcc, se, sh, cu, pipe, simd
        r[1] = t0;
        r[2] = t1;
    }
}

#include <map>
template <int OP> void run(const char* name, int insts_per_iter_x8, float* out, unsigned long long* cyc)
{
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        int blocks = 256 * w;   // 256-thread blocks: one wave per SIMD each; w blocks per CU
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, ITERS);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, ITERS);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 4 * 3);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        struct S { unsigned long long t0 = ~0ull, t1 = 0; int waves = 0; };
        std::map<unsigned long long, S> simd;
        for (size_t i = 0; i < h.size(); i += 3) {
            S& s = simd[h[i]];
            s.t0 = std::min(s.t0, h[i + 1]); s.t1 = std::max(s.t1, h[i + 2]); s.waves += 1;
        }
        const double units = (double)ITERS / 8.0 * insts_per_iter_x8;   // instructions (or rounds) per wave
        std::vector<double> rate, occ;
        for (auto& kv : simd) { rate.push_back((double)(kv.second.t1 - kv.second.t0) / (kv.second.waves * units)); occ.push_back(kv.second.waves); }
        std::sort(rate.begin(), rate.end()); std::sort(occ.begin(), occ.end());
        printf("%-28s blocks/CU %d: %.3f ms, %zu SIMDs seen, waves/SIMD min %.0f median %.0f max %.0f -> cycles per unit per SIMD: "
               "min %.2f median %.2f max %.2f\n", name, w, ms, simd.size(), occ.front(), occ[occ.size() / 2], occ.back(),
               rate.front(), rate[rate.size() / 2], rate.back());
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8 * 4 * 3 * 8);
    run<0>("v_fma_f32 x8 independent", 64, out, cyc);
    run<1>("v_fma_f32 dependent chain", 64, out, cyc);
    run<2>("v_sqrt_f32 x8 independent", 64, out, cyc);
    run<3>("v_fma_f64 x4 independent", 64, out, cyc);
    run<6>("v_pk_fma_f32 x4 independent", 64, out, cyc);
    run<7>("v_add_u32 x8 chain", 64, out, cyc);
    run<4>("exact pair force (round)", 1, out, cyc);
    run<5>("fast pair force (round)", 1, out, cyc);
    return 0;
}
