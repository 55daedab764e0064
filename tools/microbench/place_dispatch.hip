// place_dispatch.hip -- the library's place_kernel, returning at its first instruction (diagnostics switch 128),
// launched standalone on 3907 x 256 threads beside a two-argument empty kernel: is its 17 us a property of the kernel?
//   hipcc -O3 -std=c++17 -DPEDONI_DIAGNOSTICS -I include -I pedoni_amd/csrc --offload-arch=gfx950 ...
#include <hip/hip_runtime.h>
#include "kernels.hpp"
#include <cstdio>
#include <vector>
using namespace pedoni;
__global__ void tiny(uint32_t* p, uint32_t n) { if (n == 0xffffffffu) p[0] = 1; }
int main()
{
    uint32_t* buf; hipMalloc((void**)&buf, 1 << 24); hipMemset(buf, 0, 1 << 24);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    GridView grid{1.4f, 715, 715};
    BandView band{0, 715, 0};
    SoA soa{}; soa.pos_in = (const float2*)buf; soa.velx_in = (const float4*)buf; soa.dest_in = buf;
    soa.pos_out = (float2*)buf; soa.velx_out = (float4*)buf; soa.dest_out = buf; soa.skey_out = buf;
    auto run = [&](auto launch, const char* name) {
        float tot = 0;
        for (int i = 0; i < 25; ++i) {
            hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 5) tot += ms;
        }
        std::printf("%-52s: %6.2f us\n", name, tot * 1e3 / 20);
    };
    const dim3 g(3907), blk(256);
    run([&] { hipLaunchKernelGGL(tiny, g, blk, 0, 0, buf, 1u); }, "two-argument empty kernel");
    run([&] { hipLaunchKernelGGL(place_kernel, g, blk, 0, 0, buf, 0u, 1000000u, grid, band, buf, buf, (SortFlags*)buf, 0u, buf, soa, buf,
                                 (HaloIn*)nullptr, buf, 0, 715, buf, (uint32_t*)nullptr, (uint32_t*)nullptr, 128u); },
        "place_kernel returning at once (switch 128)");
    run([&] { hipLaunchKernelGGL(place_kernel, dim3(977), dim3(1024), 0, 0, buf, 0u, 1000000u, grid, band, buf, buf, (SortFlags*)buf, 0u, buf, soa, buf,
                                 (HaloIn*)nullptr, buf, 0, 715, buf, (uint32_t*)nullptr, (uint32_t*)nullptr, 128u); },
        "the same, 977 x 1024 threads");
    // the same WITHOUT a host synchronisation per launch: 200 x [tiny, place(128)] queued at once on a non-blocking
    // stream, every launch inside its own event pair (read afterwards)
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int reps = 200;
    std::vector<hipEvent_t> ev(4 * reps);
    for (auto& e : ev) hipEventCreate(&e);
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(ev[4 * i], st);
        hipLaunchKernelGGL(tiny, g, blk, 0, st, buf, 1u);
        hipEventRecord(ev[4 * i + 1], st);
        hipEventRecord(ev[4 * i + 2], st);
        hipLaunchKernelGGL(place_kernel, g, blk, 0, st, buf, 0u, 1000000u, grid, band, buf, buf, (SortFlags*)buf, 0u, buf, soa, buf,
                           (HaloIn*)nullptr, buf, 0, 715, buf, (uint32_t*)nullptr, (uint32_t*)nullptr, 128u);
        hipEventRecord(ev[4 * i + 3], st);
    }
    hipStreamSynchronize(st);
    float t_tiny = 0, t_place = 0;
    for (int i = 20; i < reps; ++i) {
        float ms; hipEventElapsedTime(&ms, ev[4 * i], ev[4 * i + 1]); t_tiny += ms;
        hipEventElapsedTime(&ms, ev[4 * i + 2], ev[4 * i + 3]); t_place += ms;
    }
    std::printf("queued without host syncs, non-blocking stream: tiny %.2f us, place_kernel(128) %.2f us\n",
                t_tiny * 1e3 / (reps - 20), t_place * 1e3 / (reps - 20));
    return 0;
}
