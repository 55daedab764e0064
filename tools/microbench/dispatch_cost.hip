// dispatch_cost.hip -- why does an EMPTY place_kernel take 18 us?  Empty kernels of 3907 x 256 threads, timed by event
// pairs, with: no LDS / 4 bytes of LDS; few / ~200 bytes of arguments; and after a preceding small kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct Big { const void* p[12]; uint32_t u[24]; };
__global__ void e_plain(uint32_t* p, uint32_t n) { if (n == 0xffffffffu) p[0] = 1; }
__global__ void e_lds(uint32_t* p, uint32_t n) { __shared__ uint32_t s; if (n == 0xffffffffu) { s = 1; __syncthreads(); p[0] = s; } }
__global__ void e_args(Big b, uint32_t* p, uint32_t n) { if (n == 0xffffffffu) p[0] = b.u[3]; }
__global__ void e_both(Big b, uint32_t* p, uint32_t n) { __shared__ uint32_t s; if (n == 0xffffffffu) { s = b.u[1]; __syncthreads(); p[0] = s; } }
__global__ void small_k(uint32_t* c, uint32_t n) { uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; if (j < n) c[j] = 0; }
int main()
{
    uint32_t* s; hipMalloc((void**)&s, 600000 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    Big big{};
    auto run = [&](auto launch, const char* name) {
        float tot = 0;
        for (int i = 0; i < 25; ++i) {
            hipLaunchKernelGGL(small_k, dim3(715), dim3(256), 0, 0, s, 511225u);
            hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 5) tot += ms;
        }
        std::printf("%-44s: %6.2f us\n", name, tot * 1e3 / 20);
    };
    const dim3 g(3907), blk(256);
    run([&] { hipLaunchKernelGGL(e_plain, g, blk, 0, 0, s, 1u); }, "empty, no LDS, 2 arguments");
    run([&] { hipLaunchKernelGGL(e_lds, g, blk, 0, 0, s, 1u); }, "empty, 4 B of LDS");
    run([&] { hipLaunchKernelGGL(e_args, g, blk, 0, 0, big, s, 1u); }, "empty, 200 B of arguments");
    run([&] { hipLaunchKernelGGL(e_both, g, blk, 0, 0, big, s, 1u); }, "empty, 4 B of LDS + 200 B of arguments");
    run([&] { hipLaunchKernelGGL(e_plain, dim3(15628), dim3(64), 0, 0, s, 1u); }, "empty, 15628 x 64 threads");
    return 0;
}
