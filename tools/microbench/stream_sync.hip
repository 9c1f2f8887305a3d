// Cost of ordering work across two streams on one GPU: event record + wait against stream value write + wait.
// Build: make -C tools/microbench stream_sync   Run: tools/microbench/stream_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_spin(int *p, int iters) { int v = 0; for (int i = 0; i < iters; i++) v += i; if (v == 123456789) *p = v; }
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int *dummy; CK(hipMalloc(&dummy, 4));
    uint32_t *sig = nullptr, *sig2 = nullptr;                  // signal memory comes in 8-byte allocations
    CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void **)&sig2, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8)); CK(hipMemset(sig2, 0, 8));
    const int N = 2000;
    hipEvent_t ev[2];
    for (int k = 0; k < 2; k++) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto t0, auto t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
    // baseline: the two kernels of an iteration on ONE stream
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        for (int i = 0; i < N; i++) { k_spin<<<1, 64, 0, a>>>(dummy, 100); k_spin<<<1, 64, 0, a>>>(dummy, 100); }
        CK(hipStreamSynchronize(a));
        if (rep) printf("one stream, 2 kernels per iteration:               %.2f us per iteration\n", us(t0, now()) / N);
    }
    // events: kernel on b, record, a waits, kernel on a, record, b waits (ping-pong)
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        for (int i = 0; i < N; i++) {
            k_spin<<<1, 64, 0, b>>>(dummy, 100);
            CK(hipEventRecord(ev[0], b)); CK(hipStreamWaitEvent(a, ev[0], 0));
            k_spin<<<1, 64, 0, a>>>(dummy, 100);
            CK(hipEventRecord(ev[1], a)); CK(hipStreamWaitEvent(b, ev[1], 0));
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        if (rep) printf("two streams, event record + wait both ways:         %.2f us per iteration\n", us(t0, now()) / N);
    }
    // stream values: the same ping-pong through two words of signal memory
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(sig, 0, 8)); CK(hipMemset(sig2, 0, 8));
        auto t0 = now();
        for (int i = 0; i < N; i++) {
            k_spin<<<1, 64, 0, b>>>(dummy, 100);
            CK(hipStreamWriteValue32(b, sig, (uint32_t)(i + 1), 0));
            CK(hipStreamWaitValue32(a, sig, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
            k_spin<<<1, 64, 0, a>>>(dummy, 100);
            CK(hipStreamWriteValue32(a, sig2, (uint32_t)(i + 1), 0));
            CK(hipStreamWaitValue32(b, sig2, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        if (rep) printf("two streams, value write + wait both ways:          %.2f us per iteration\n", us(t0, now()) / N);
    }
    // one-way only (producer -> consumer), as a load stream feeding the main stream with a deep ring
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(sig, 0, 8)); CK(hipMemset(sig2, 0, 8));
        auto t0 = now();
        for (int i = 0; i < N; i++) {
            k_spin<<<1, 64, 0, b>>>(dummy, 100);
            CK(hipStreamWriteValue32(b, sig, (uint32_t)(i + 1), 0));
            CK(hipStreamWaitValue32(a, sig, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
            k_spin<<<1, 64, 0, a>>>(dummy, 100);
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        if (rep) printf("two streams, value write + wait one way:            %.2f us per iteration\n", us(t0, now()) / N);
    }
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        for (int i = 0; i < N; i++) {
            k_spin<<<1, 64, 0, b>>>(dummy, 100);
            CK(hipEventRecord(ev[0], b)); CK(hipStreamWaitEvent(a, ev[0], 0));
            k_spin<<<1, 64, 0, a>>>(dummy, 100);
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        if (rep) printf("two streams, event record + wait one way:           %.2f us per iteration\n", us(t0, now()) / N);
    }
    return 0;
}
