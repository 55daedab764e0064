// move_records.hip -- the floor of place_kernel's data movement: N records of 28 bytes (float2 + float4 + u32) read at j and
// written at j (identity placement), plus a 4-byte key read and a 4-byte packed-cell write, as 256-thread workgroups with
// one record per thread -- and the same with 2 / 4 records per thread.   hipcc -O3 --offload-arch=gfx950 ... && ./move_records
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int PER>
__global__ void mv(const uint32_t* __restrict__ key, const float2* __restrict__ p, const float4* __restrict__ v, const uint32_t* __restrict__ d,
                   float2* __restrict__ po, float4* __restrict__ vo, uint32_t* __restrict__ dout, uint32_t* __restrict__ sk, uint32_t n)
{
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t j = (blockIdx.x * PER + k) * blockDim.x + threadIdx.x;
        if (j >= n) return;
        const uint32_t c = key[j];
        if (c == 0xffffffffu) continue;
        po[j] = p[j]; vo[j] = v[j]; dout[j] = d[j]; sk[j] = c;
    }
}
__global__ void empty_kernel(const uint32_t* __restrict__ key, const float2* __restrict__, const float4* __restrict__, const uint32_t* __restrict__,
                             float2* __restrict__, float4* __restrict__, uint32_t* __restrict__, uint32_t* __restrict__ sk, uint32_t n)
{
    if (n == 0xffffffffu) sk[0] = key[0];
}
// many scattered atomics first (what the force kernel leaves behind), then the move: does the move pay for them?
__global__ void atomics_kernel(uint32_t* __restrict__ counts, uint32_t n)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) atomicAdd(&counts[(j * 2654435761u) % 511225u], 1u);
}
int main()
{
    const uint32_t n = 1000000;
    uint32_t *key, *d, *dout, *sk; float2 *p, *po; float4 *v, *vo;
    hipMalloc((void**)&key, n * 4); hipMalloc((void**)&d, n * 4); hipMalloc((void**)&dout, n * 4); hipMalloc((void**)&sk, n * 4);
    hipMalloc((void**)&p, n * 8); hipMalloc((void**)&po, n * 8); hipMalloc((void**)&v, n * 16); hipMalloc((void**)&vo, n * 16);
    hipMemset(key, 0, n * 4); hipMemset(p, 0, n * 8); hipMemset(v, 0, n * 16); hipMemset(d, 0, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, int per, const char* name) {
        const dim3 grid((n + 256 * per - 1) / (256 * per));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, key, p, v, d, po, vo, dout, sk, n);
        hipEventRecord(a, 0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, key, p, v, d, po, vo, dout, sk, n);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::printf("%s: %.2f us per launch, %.2f TB/s (64 MB moved)\n", name, ms * 1e3 / 50, 64e6 / (ms * 1e-3 / 50) / 1e12);
    };
    run(mv<1>, 1, "1 record per thread "); run(mv<2>, 2, "2 records per thread"); run(mv<4>, 4, "4 records per thread");
    run(empty_kernel, 1, "empty kernel, 3907 workgroups");
    {   // alternate: 1e6 scattered atomics, then the move (each timed by its own event pair, as the library's profile does)
        uint32_t* counts; hipMalloc((void**)&counts, 511225 * 4 + 64); hipMemset(counts, 0, 511225 * 4);
        hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
        float t_at = 0, t_mv = 0;
        for (int i = 0; i < 30; ++i) {
            hipEventRecord(e[0], 0);
            hipLaunchKernelGGL(atomics_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, counts, n);
            hipEventRecord(e[1], 0);
            hipEventRecord(e[2], 0);
            hipLaunchKernelGGL(mv<1>, dim3((n + 255) / 256), dim3(256), 0, 0, key, p, v, d, po, vo, dout, sk, n);
            hipEventRecord(e[3], 0);
            hipEventSynchronize(e[3]);
            float a1, a2; hipEventElapsedTime(&a1, e[0], e[1]); hipEventElapsedTime(&a2, e[2], e[3]);
            if (i >= 5) { t_at += a1; t_mv += a2; }
        }
        std::printf("event-pair timed, alternating: 1e6 scattered atomics %.2f us, the move right after them %.2f us\n", t_at * 1e3 / 25, t_mv * 1e3 / 25);
    }
    return 0;
}
