// stream_wait_value.hip -- can a kernel in flight release work on ANOTHER stream?  (The overlapped band tick wants
// the exchange to start when the edge rows' tiles of the force kernel are done, not when the whole launch is.)
// Stream 1: a kernel of 3907 x 256 threads that spins ~80 us per workgroup wave; the workgroups with blockIdx < 64
// count themselves in when done and the last of them stores the tick number to a word of signal memory
// (hipExtMallocWithFlags(hipMallocSignalMemory)).  Stream 2: hipStreamWaitValue32(word >= tick), then a tiny kernel
// that stamps the clock.  Printed: when the stamp kernel ran relative to the long kernel's first and last wave.
// Every wait is satisfiable by construction (the long kernel always stores the value); the host gives up after 5 s.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void long_kernel(uint32_t* counter, uint32_t* flag, uint32_t tick, uint32_t first_n, unsigned long long* stamps,
                            uint32_t spin)
{
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = t0;
    while (wall_clock64() - t0 < spin) {}            // (100 MHz clock: spin = 800 -> 8 us per workgroup)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x < first_n) {
            __threadfence();
            if (atomicAdd(counter, 1u) + 1u == first_n) {
                *counter = 0;
                stamps[1] = wall_clock64();
                __hip_atomic_store(flag, tick, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        atomicMax(&stamps[2], wall_clock64());
    }
}
__global__ void stamp_kernel(unsigned long long* stamps) { if (threadIdx.x == 0) stamps[3] = wall_clock64(); }

int main()
{
    int can = 0;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    uint32_t *counter, *flag;
    unsigned long long *stamps, h[4];
    CHECK(hipMalloc((void**)&counter, 4));
    CHECK(hipMemset(counter, 0, 4));
    CHECK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    CHECK(hipMalloc((void**)&stamps, 32));
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, -1));
    CHECK(hipStreamWriteValue32(s2, flag, 0, 0));
    CHECK(hipStreamSynchronize(s2));
    hipEvent_t done;
    CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    for (uint32_t tick = 1; tick <= 12; ++tick) {
        CHECK(hipMemsetAsync(stamps, 0, 32, s1));
        CHECK(hipStreamSynchronize(s1));
        // 3907 workgroups, 7 resident per CU on 256 CUs = 2.2 rounds of `spin`
        hipLaunchKernelGGL(long_kernel, dim3(3907), dim3(256), 0, s1, counter, flag, tick, 64u, stamps, 3000u);
        CHECK(hipStreamWaitValue32(s2, flag, tick, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s2, stamps);
        CHECK(hipEventRecord(done, s2));
        const auto t0 = std::chrono::steady_clock::now();
        while (hipEventQuery(done) == hipErrorNotReady) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { std::printf("tick %u: the wait never ended\n", tick); return 2; }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        CHECK(hipStreamSynchronize(s1));
        CHECK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
        if (tick > 2)
            std::printf("tick %2u: first 64 workgroups done at +%6.1f us, stamp kernel on the other stream at +%6.1f us, "
                        "long kernel's last wave at +%6.1f us\n", tick, (h[1] - h[0]) / 100.0, (h[3] - h[0]) / 100.0,
                        (h[2] - h[0]) / 100.0);
    }
    return 0;
}
