// rccl_self_exchange.hip -- what does ONE neighbour exchange of the band tick cost, host side and device side, before any
// wire?  One rank (a 1-GPU box), real librccl: a grouped ncclSend + ncclRecv pair addressed to itself, twice per "tick"
// (the down list and the up list: 129 KB each at the bench's halo capacity), on a stream of its own, 200 times.
// Printed: host time per grouped exchange (the enqueue cost the tick's host thread pays every tick) and the device time
// from the first RCCL kernel's start to the last one's end (event pair), idle stream.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/rse tools/microbench/rccl_self_exchange.hip -L/opt/rocm/lib -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::printf("%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
int main()
{
    const size_t words = 6 + 2304 * 14;            // header + 2304 records of 56 bytes
    uint32_t *send, *recv;
    CHECK(hipMalloc((void**)&send, 2 * words * 4));
    CHECK(hipMalloc((void**)&recv, 2 * words * 4));
    CHECK(hipMemset(send, 1, 2 * words * 4));
    ncclUniqueId id;
    NCHECK(ncclGetUniqueId(&id));
    ncclComm_t comm;
    NCHECK(ncclCommInitRank(&comm, 1, id, 0));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto exchange = [&]() -> int {
        NCHECK(ncclGroupStart());
        NCHECK(ncclSend(send, words, ncclUint32, 0, comm, st));
        NCHECK(ncclRecv(recv, words, ncclUint32, 0, comm, st));
        NCHECK(ncclSend(send + words, words, ncclUint32, 0, comm, st));
        NCHECK(ncclRecv(recv + words, words, ncclUint32, 0, comm, st));
        NCHECK(ncclGroupEnd());
        return 0;
    };
    for (int i = 0; i < 20; ++i) if (exchange()) return 1;
    CHECK(hipStreamSynchronize(st));
    double host_us = 0, dev_us = 0;
    const int reps = 200;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(a, st));
        const auto t0 = std::chrono::steady_clock::now();
        if (exchange()) return 1;
        host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        CHECK(hipEventRecord(b, st));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        dev_us += ms * 1e3;
    }
    std::printf("grouped 2 x (ncclSend + ncclRecv) of %zu bytes each, one rank, self-addressed: host %.1f us per exchange, "
                "device %.1f us (event pair around it, idle stream)\n", words * 4, host_us / reps, dev_us / reps);
    // back to back, no sync: the host's sustained rate
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) if (exchange()) return 1;
    const double enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CHECK(hipStreamSynchronize(st));
    const double all = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("200 exchanges back to back: host enqueue %.1f us each, %.1f us each until the stream drained\n", enq / reps, all / reps);
    ncclCommDestroy(comm);
    return 0;
}
