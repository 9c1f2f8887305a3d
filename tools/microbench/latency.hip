// Latency microbenchmarks on one MI355X: what a dependent memory round trip / cross-XCD hand-off /
// kernel boundary costs.  Results steer the fusion decisions in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1.0f; }

// one wave: chain of `n` dependent loads of kind `mode`; reports cycles (wall_clock64 = 100 MHz const clock)
__global__ void k_chase(unsigned int *buf, int n, int stride, int mode, long long *out) {
    if (threadIdx.x != 0) return;
    unsigned int idx = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        unsigned int v;
        if (mode == 0) v = buf[idx];                                                                            // plain
        else if (mode == 1) v = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // sc1
        else v = __hip_atomic_fetch_add(buf + idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);              // atomic rtn
        idx = (v + (unsigned)(i + 1) * (unsigned)stride);
    }
    long long t1 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = idx;
}

// store + wait completion
__global__ void k_store(unsigned int *buf, int n, int mode, long long *out) {
    if (threadIdx.x != 0) return;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        if (mode == 0) buf[i * 64] = i; else __hip_atomic_store(buf + i * 64, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }
    long long t1 = wall_clock64();
    out[0] = t1 - t0;
}

// G workgroups arrive at a counter; the last one records when (relative to the earliest start)
__global__ void k_arrive(unsigned int *ticket, long long *tmin, long long *out, int levels) {
    __shared__ int last;
    long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin((unsigned long long *)tmin, (unsigned long long)t0);
        unsigned int old;
        int l = 0;
        if (levels == 2) {
            int g = blockIdx.x / 16, ng = (gridDim.x + 15) / 16;
            unsigned expect = min(16u, gridDim.x - g * 16u);
            old = __hip_atomic_fetch_add(ticket + 32 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == expect - 1u) { old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); l = (old == (unsigned)ng - 1u); }
        } else {
            old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l = (old == gridDim.x - 1);
        }
        last = l;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        long long t1 = wall_clock64();
        out[0] = t1 - __hip_atomic_load((long long *)tmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < 40; k++) ticket[32 * k] = 0;
    }
}

int main() {
    unsigned int *buf; long long *out; long long *tmin; unsigned int *ticket;
    size_t N = 64u << 20;
    CK(hipMalloc(&buf, N * 4)); CK(hipMemset(buf, 0, N * 4));
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&tmin, 8)); CK(hipMalloc(&ticket, 4096 * 4)); CK(hipMemset(ticket, 0, 4096 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    long long h[2];
    const double ns_per_tick = 10.0;   // wall_clock64: 100 MHz
    const char *names[3] = {"plain load", "sc1 (agent) load", "agent atomic rtn"};
    for (int mode = 0; mode < 3; mode++)
        for (int stride : {0, 4099 * 32}) {   // same line (cache hit after first) vs a new far line every time
            int n = 200;
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s, buf, n, stride, mode, out);
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s, buf, n, stride, mode, out);
            CK(hipMemcpyAsync(h, out, 16, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            printf("%-20s %-10s : %7.1f ns per dependent access\n", names[mode], stride ? "far lines" : "same line", h[0] * ns_per_tick / n);
        }
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, s, buf, 100, mode, out);
        CK(hipMemcpyAsync(h, out, 16, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        printf("%-31s : %7.1f ns per store+wait\n", mode ? "agent-scope store + vmcnt(0)" : "plain store + vmcnt(0)", h[0] * ns_per_tick / 100);
    }
    for (int levels = 1; levels <= 2; levels++)
        for (int G : {64, 300, 512, 2048}) {
            double best = 1e30;
            for (int rep = 0; rep < 10; rep++) {
                long long big = 0x7fffffffffffffffLL;
                CK(hipMemcpyAsync(tmin, &big, 8, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_arrive, dim3(G), dim3(512), 0, s, ticket, tmin, out, levels);
                CK(hipMemcpyAsync(h, out, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
                best = std::min(best, h[0] * ns_per_tick);
            }
            printf("arrival counter, %d level(s), %4d WGs x 512 thr : first start -> last arrival %8.1f ns\n", levels, G, best);
        }
    // kernel boundary: N dependent empty kernels back to back
    for (int G : {1, 300, 1200}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, s);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("back-to-back empty kernels, grid %4d x 256 : %7.2f us per launch\n", G, ms * 1000 / 200);
    }
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(300), dim3(256), 0, s);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipGraph of 200 empty kernels, grid 300 x 256 : %7.2f us per kernel node\n", ms * 1000 / 200);
    }
    return 0;
}
