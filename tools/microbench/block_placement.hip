// block_placement.hip -- where does the dispatcher put the workgroups of a SMALL grid?
// K workgroups of 256 threads, each with `lds` bytes of LDS, spin ~20 us and record the CU they ran on
// (XCC_ID, SE, SH, CU from the hardware registers) and when they started.  Prints workgroups per CU.
//   hipcc -O3 --offload-arch=gfx950 -o block_placement block_placement.hip && ./block_placement 782 28000
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, unsigned long long* t0, int spin)
{
    extern __shared__ unsigned char dyn[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long start = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; t0[blockIdx.x] = start; dyn[0] = 1; }
    while (__builtin_amdgcn_s_memrealtime() - start < (unsigned long long)spin) {}     // 100 MHz ticks
}

int main(int argc, char** argv)
{
    const int k = argc > 1 ? std::atoi(argv[1]) : 782;
    const int lds = argc > 2 ? std::atoi(argv[2]) : 28000;
    unsigned* out; unsigned long long* t0;
    hipMalloc((void**)&out, k * 2 * sizeof(unsigned)); hipMalloc((void**)&t0, k * sizeof(unsigned long long));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(k), dim3(256), lds, 0, out, t0, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(k * 2); std::vector<unsigned long long> t(k);
    hipMemcpy(h.data(), out, k * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), t0, k * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu; std::map<unsigned, int> per_xcc;
    unsigned long long tmin = *std::min_element(t.begin(), t.end()), tmax = *std::max_element(t.begin(), t.end());
    for (int b = 0; b < k; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu] += 1;
        per_xcc[xcc] += 1;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second] += 1;
    std::printf("%d workgroups x %d B LDS: %zu distinct CUs used; workgroups per CU -> number of CUs:", k, lds, per_cu.size());
    for (auto& kv : hist) std::printf("  %d: %d", kv.first, kv.second);
    std::printf("\n  per XCC:");
    for (auto& kv : per_xcc) std::printf(" %u:%d", kv.first, kv.second);
    std::printf("\n  start spread: %.2f us (100 MHz ticks)\n", (double)(tmax - tmin) / 100.0);
    return 0;
}
