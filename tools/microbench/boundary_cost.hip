// boundary_cost.hip -- what does a kernel pay for what the kernels BEFORE it left behind?  An (almost) empty kernel of
// 3907 workgroups is timed with an event pair after: nothing; a kernel that writes 32 MB; one that issues 1e6 scattered
// agent-scope atomics; both; and with a small kernel in between (the tick's force -> scan -> place order).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void empty_k(uint32_t* p, uint32_t n) { if (n == 0xffffffffu) p[0] = 1; }
__global__ void write_k(float4* o, uint32_t n) { uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; if (j < n) o[j] = make_float4(j, 1, 2, 3); }
__global__ void atom_k(uint32_t* c, uint32_t n) { uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; if (j < n) atomicAdd(&c[(j * 2654435761u) % 511225u], 1u); }
__global__ void small_k(uint32_t* c, uint32_t n) { uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; if (j < n) c[j] = 0; }
int main()
{
    const uint32_t n = 1000000;
    float4* o; uint32_t *c, *s; hipMalloc((void**)&o, 2 * n * 16); hipMalloc((void**)&c, 600000 * 4); hipMalloc((void**)&s, 600000 * 4);
    hipMemset(c, 0, 600000 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const dim3 g((n + 255) / 256), blk(256);
    auto time_empty_after = [&](int what, bool small_between, const char* name) {
        float tot = 0;
        for (int i = 0; i < 25; ++i) {
            if (what & 1) hipLaunchKernelGGL(write_k, dim3(2 * g.x), blk, 0, 0, o, 2 * n);
            if (what & 2) hipLaunchKernelGGL(atom_k, g, blk, 0, 0, c, n);
            if (small_between) hipLaunchKernelGGL(small_k, dim3(715), blk, 0, 0, s, 511225u);
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(empty_k, g, blk, 0, 0, s, n);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 5) tot += ms;
        }
        std::printf("empty kernel after %-52s: %6.2f us\n", name, tot * 1e3 / 20);
    };
    time_empty_after(0, false, "nothing");
    time_empty_after(1, false, "a kernel that wrote 32 MB");
    time_empty_after(2, false, "1e6 scattered agent-scope atomics");
    time_empty_after(3, false, "both");
    time_empty_after(3, true, "both, then a small kernel (715 workgroups, 2 MB)");
    time_empty_after(2, true, "the atomics, then the small kernel");
    time_empty_after(1, true, "the 32 MB writer, then the small kernel");
    return 0;
}
