// scalar_stages.hip -- what a launch pays per DEPENDENT scalar-load stage before its waves can leave: empty kernels of
// 3907 x 256 threads whose every wave first walks a chain of k dependent scalar loads (kernarg -> global word -> global
// word ...), k = 0 .. 4; and the same with many kernel arguments read in ONE stage.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int K> __global__ void chain(const uint64_t* start, uint32_t* out, uint32_t never)
{
    const uint64_t* p = start;
#pragma unroll
    for (int k = 0; k < K; ++k) p = reinterpret_cast<const uint64_t*>(*p);     // uniform: scalar loads
    if (reinterpret_cast<uint64_t>(p) == never) out[0] = 1;
}
struct Many { uint64_t a[24]; };
__global__ void many_args(Many m, uint32_t* out, uint32_t never)
{
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 24; ++k) s += m.a[k];
    if (s == never) out[0] = 1;
}
int main()
{
    uint64_t* cells; hipMalloc((void**)&cells, 8 * 4096);
    uint64_t h[8 * 512] = {};
    for (int k = 0; k < 7; ++k) h[k * 512] = reinterpret_cast<uint64_t>(cells + (k + 1) * 512);    // one 4-KB page apart
    h[7 * 512] = reinterpret_cast<uint64_t>(cells);
    hipMemcpy(cells, h, sizeof h, hipMemcpyHostToDevice);
    uint32_t* out; hipMalloc((void**)&out, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const dim3 g(3907), blk(256);
    auto run = [&](auto launch, const char* name) {
        float tot = 0;
        for (int i = 0; i < 25; ++i) {
            hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 5) tot += ms;
        }
        std::printf("%-40s: %6.2f us\n", name, tot * 1e3 / 20);
    };
    run([&] { hipLaunchKernelGGL(chain<0>, g, blk, 0, 0, cells, out, 1u); }, "0 dependent global scalar loads");
    run([&] { hipLaunchKernelGGL(chain<1>, g, blk, 0, 0, cells, out, 1u); }, "1");
    run([&] { hipLaunchKernelGGL(chain<2>, g, blk, 0, 0, cells, out, 1u); }, "2");
    run([&] { hipLaunchKernelGGL(chain<3>, g, blk, 0, 0, cells, out, 1u); }, "3");
    run([&] { hipLaunchKernelGGL(chain<4>, g, blk, 0, 0, cells, out, 1u); }, "4");
    Many m{};
    run([&] { hipLaunchKernelGGL(many_args, g, blk, 0, 0, m, out, 1u); }, "24 x 8-byte arguments, one stage");
    return 0;
}
