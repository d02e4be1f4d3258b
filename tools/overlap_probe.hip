// Does a long-running kernel on one stream slow down a chain of short dependent kernels on another?
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe && tools/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void tiny(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }

// spins for about `cycles` wall-clock ticks (100 MHz) per workgroup; `rounds` workgroup generations via the grid size
__global__ void longk(float* p, long long ticks) {
    const long long t0 = wall_clock64();
    float x = p[blockIdx.x];
    while (wall_clock64() - t0 < ticks) x = x * 1.0001f + 0.5f;
    p[blockIdx.x] = x;
}

static float chain_ms(hipStream_t a, float* buf, int n, int iters, int grid_tiny) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tiny, dim3(grid_tiny), dim3(256), 0, a, buf, n);
    CK(hipEventRecord(e1, a)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

// background kernel that keeps writing: mode 0 plain stores, 1 nontemporal stores, 2 atomic adds, 3 loads only
__global__ __launch_bounds__(256) void writer(float* buf, size_t n, long long ticks, int mode) {
    const long long t0 = wall_clock64();
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 0) buf[i] = acc;
        else if (mode == 1) __builtin_nontemporal_store(acc, buf + i);
        else if (mode == 2) atomicAdd(buf + i, 1.0f);
        else acc += buf[i];
        acc += 1.f;
        i += stride; if (i >= n) i -= n;
    }
    if (mode == 3 && acc == 12345.f) buf[0] = acc;
}

__global__ void stamp(unsigned long long* out) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = wall_clock64(); }

// the same experiment as ONE hipGraph: [long kernels on stream b] beside [stamp, chain of tiny kernels, stamp on stream a]
static void graph_case(hipStream_t a, hipStream_t b, float* buf, int n, float* buf2, int wgs, int threads, long long ticks,
                       int launches, int iters, const char* name) {
    unsigned long long* st; CK(hipMalloc(&st, 16));
    hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, a)); CK(hipStreamWaitEvent(b, fork, 0));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(longk, dim3(wgs), dim3(threads), 0, b, buf2, ticks);
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, st);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tiny, dim3(128), dim3(256), 0, a, buf, n);
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, st + 1);
    CK(hipEventRecord(join, b)); CK(hipStreamWaitEvent(a, join, 0));
    CK(hipStreamEndCapture(a, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, a)); CK(hipStreamSynchronize(a));
    CK(hipGraphLaunch(ge, a)); CK(hipStreamSynchronize(a));
    unsigned long long h[2]; CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
    printf("GRAPH beside [%s]: %.2f us per tiny kernel (chain took %.2f ms)\n", name, (h[1] - h[0]) * 0.01 / iters, (h[1] - h[0]) * 1e-5);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    const int n = 64 * 512; float *buf, *buf2;
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&buf2, 1 << 20)); CK(hipMemset(buf, 0, n * 4)); CK(hipMemset(buf2, 0, 1 << 20));
    const int iters = 200;
    chain_ms(a, buf, n, 50, 128);
    printf("chain of %d tiny kernels alone: %.2f us per kernel\n", iters, chain_ms(a, buf, n, iters, 128) * 1e3 / iters);
    struct Case { const char* name; int wgs; int threads; long long ticks; int launches; };
    // total background duration ~3 ms in each case
    const Case cases[] = {{"1 long kernel, 16 workgroups, 3 ms", 16, 256, 300000, 1},
                          {"1 long kernel, 256 workgroups (1 per CU), 3 ms", 256, 256, 300000, 1},
                          {"1 long kernel, 2048 workgroups x 64 threads, each 0.4 ms (8 generations)", 2048 * 8, 64, 40000, 1},
                          {"30 kernels of 100 us, 256 workgroups", 256, 256, 10000, 30},
                          {"150 kernels of 20 us, 256 workgroups", 256, 256, 2000, 150}};
    for (const Case& c : cases) {
        for (int l = 0; l < c.launches; ++l) hipLaunchKernelGGL(longk, dim3(c.wgs), dim3(c.threads), 0, b, buf2, c.ticks);
        const float ms = chain_ms(a, buf, n, iters, 128);
        CK(hipStreamSynchronize(b));
        printf("beside [%s]: %.2f us per tiny kernel (chain took %.2f ms)\n", c.name, ms * 1e3 / iters, ms);
        graph_case(a, b, buf, n, buf2, c.wgs, c.threads, c.ticks, c.launches, iters, c.name);
    }
    {
        const size_t nw = (size_t)64 << 20;      // 256 MB of floats
        float* big; CK(hipMalloc(&big, nw * 4)); CK(hipMemset(big, 0, nw * 4));
        const char* names[4] = {"plain stores", "nontemporal stores", "atomic adds", "loads only"};
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(writer, dim3(512), dim3(256), 0, b, big, nw, 300000, mode);
            const float ms = chain_ms(a, buf, n, iters, 128);
            CK(hipStreamSynchronize(b));
            printf("beside [1 kernel of 512 workgroups streaming %s over 256 MB for 3 ms]: %.2f us per tiny kernel\n", names[mode], ms * 1e3 / iters);
        }
    }
    return 0;
}
