// gather_traffic.hip -- what rocprofv3's FETCH_SIZE reports for reads of KNOWN size and shape
// (the 2x read correction of MI355X_MICROARCH.md is calibrated on 16-B-per-lane
// streams; the force kernel's reads are 4 / 8 / 16-byte gathers).  Every kernel reads N elements of a
// 1 GiB array (far larger than L2 + Infinity Cache, so every distinct line comes from HBM):
//   stream16   lane i reads float4 i                      -> 16 N bytes, in whole 128-B lines
//   gather4/8/16  lane i reads 4 / 8 / 16 bytes at a random, 16-B-aligned place (index array read
//                 coalesced: 4 N bytes more)              -> one 64-B sector (one 128-B line) each
// Run:  hipcc -O3 --offload-arch=gfx950 -o gather_traffic gather_traffic.hip
//       rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./gather_traffic      (tools/gather_traffic.sh)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stream16(const float4* __restrict__ a, uint32_t n, float* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = a[i];
    if (v.x + v.y + v.z + v.w == 12345.0f) out[0] = 1.0f;
}
template <typename T> __global__ void gather(const char* __restrict__ a, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T v = *reinterpret_cast<const T*>(a + (size_t)idx[i] * 16u);
    const float* f = reinterpret_cast<const float*>(&v);
    float s = 0.0f;
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) s += f[k];
    if (s == 12345.0f) out[0] = 1.0f;
}

int main()
{
    const size_t bytes = 1ull << 30;             // 1 GiB
    const uint32_t n = 1u << 24;                 // 16.8 M elements per launch
    char* a = nullptr; uint32_t* idx = nullptr; float* out = nullptr;
    CHECK(hipMalloc((void**)&a, bytes)); CHECK(hipMemset(a, 0, bytes));
    CHECK(hipMalloc((void**)&idx, n * sizeof(uint32_t))); CHECK(hipMalloc((void**)&out, 64));
    std::vector<uint32_t> h(n);
    std::mt19937 rng(7);
    std::uniform_int_distribution<uint32_t> pick(0, (uint32_t)(bytes / 16) - 1);
    for (auto& x : h) x = pick(rng);
    CHECK(hipMemcpy(idx, h.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    const dim3 block(256), grid(n / 256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const float4*)a, n, out);
        hipLaunchKernelGGL(gather<float>, grid, block, 0, 0, a, idx, n, out);
        hipLaunchKernelGGL(gather<float2>, grid, block, 0, 0, a, idx, n, out);
        hipLaunchKernelGGL(gather<float4>, grid, block, 0, 0, a, idx, n, out);
    }
    CHECK(hipDeviceSynchronize());
    std::printf("n = %u elements per launch; stream16 = %.1f MB; gathers: %.1f MB of index + one sector per element (%.1f MB at 64 B, %.1f MB at 128 B)\n",
                n, 16.0 * n / 1e6, 4.0 * n / 1e6, 64.0 * n / 1e6, 128.0 * n / 1e6);
    return 0;
}
