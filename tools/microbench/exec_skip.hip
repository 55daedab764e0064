// Does a VALU instruction with only the low N lanes of a wave64 enabled issue faster on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int active, int iters)
{
    int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b = 0.999f, c = 1e-4f, d = a * 0.5f, e = a * 0.25f, f = a * 0.125f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c);
            e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c);
            a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c);
            e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f;
}
int main()
{
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int active : {64, 48, 33, 32, 17, 16, 8, 1}) {
        k<<<4096, 256>>>(out, active, 2000); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<4096, 256>>>(out, active, 20000); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("active lanes %2d: %.3f ms\n", active, ms);
    }
    return 0;
}
