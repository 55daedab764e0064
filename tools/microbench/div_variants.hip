// div_variants.hip -- is ONE residual correction of the quotient enough?  device_math.hpp's div_core
// replays hipcc's IEEE division: y = refined reciprocal, q0 = n * y, then TWO corrections
// q <- fma(fma(-d, q, n), y, q).  Every operation scales exactly with powers of two (no operand
// or residual under- or overflows inside pair_force_hot's folded range test), so the result's
// significand depends on the two significands only: 2^23 x 2^23 pairs, all of them checked here.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o /tmp/div_variants tools/microbench/div_variants.hip && /tmp/div_variants
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void sweep(uint32_t n_lo, uint32_t n_hi, unsigned long long* bad, uint32_t* example)
{
    const uint32_t md = blockIdx.x * blockDim.x + threadIdx.x;          // significand of d, 0 .. 2^23 - 1
    const float d = __uint_as_float(0x3F800000u | md);                  // [1, 2)
    float y = __builtin_amdgcn_rcpf(d);
    y = __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);              // recip_refined
    unsigned long long b = 0, c = 0;
    uint32_t first = 0xffffffffu;
    for (uint32_t mn = n_lo; mn < n_hi; ++mn) {
        const float n = __uint_as_float(0x3F800000u | mn);
        const float q0 = n * y;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), y, q0);
        const float q2 = __builtin_fmaf(__builtin_fmaf(-d, q1, n), y, q1);
        c += __float_as_uint(q0) != __float_as_uint(q2);                 // control: the FIRST correction matters
        if (__float_as_uint(q1) != __float_as_uint(q2)) { b += 1; if (first == 0xffffffffu) first = mn; }
    }
    if (c) atomicAdd(bad + 1, c);
    if (b) {
        atomicAdd(bad, b);
        if (atomicCAS(&example[0], 0xffffffffu, md) == 0xffffffffu) example[1] = first;
    }
}

int main()
{
    unsigned long long* d_bad; uint32_t* d_ex;
    if (hipMalloc((void**)&d_bad, 2 * sizeof *d_bad) != hipSuccess || hipMalloc((void**)&d_ex, 8) != hipSuccess) return 1;
    (void)hipMemset(d_bad, 0, 2 * sizeof *d_bad);
    (void)hipMemset(d_ex, 0xff, 8);
    const uint32_t N = 1u << 23, STEP = 1u << 16;                       // 128 launches
    for (uint32_t lo = 0; lo < N; lo += STEP) {
        hipLaunchKernelGGL(sweep, dim3(N / 256), dim3(256), 0, 0, lo, lo + STEP, d_bad, d_ex);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        if ((lo / STEP) % 16 == 15) { printf("numerator significands < %u done\n", lo + STEP); fflush(stdout); }
    }
    unsigned long long bad, both[2]; uint32_t ex[2];
    (void)hipMemcpy(both, d_bad, sizeof both, hipMemcpyDeviceToHost);
    bad = both[0];
    (void)hipMemcpy(ex, d_ex, sizeof ex, hipMemcpyDeviceToHost);
    printf("2^46 significand pairs: %llu where the second correction changes the quotient", bad);
    if (bad) printf(" (e.g. d = 0x%08x, n = 0x%08x)", 0x3F800000u | ex[0], 0x3F800000u | ex[1]);
    printf("\ncontrol: the first correction changes n * y in %llu pairs\n", both[1]);
    return 0;
}
