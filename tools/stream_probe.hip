// What can ONE launch that only reads N MB (and writes a few KB) reach on MI355X, with the data coming from
// HBM (rotation over > 2 x 256 MiB of distinct buffers) or from the Infinity Cache (the same buffer again)?
// This is the ceiling the decoder-attention launch (54 MB at config 2) is held against.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// every workgroup sums a contiguous slab; U independent 16-byte loads in flight per thread
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, size_t n4, float* out) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = blockIdx.x * per, hi = min(n4, lo + per);
    float4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
    }
    for (; i < hi; i += 256) { const float4 v = src[i]; acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w; }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
    if (s == 123456.789f) out[blockIdx.x] = s;      // keeps the loads alive, practically never stores
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* out; CK(hipMalloc(&out, 1 << 20));
    for (double mb : {27.1, 54.2, 209.6}) {
        const size_t bytes = (size_t)(mb * 1e6) / 16 * 16, n4 = bytes / 16;
        const int n_rot = (int)((2.2 * 268435456.0) / bytes) + 2;
        std::vector<float4*> bufs(n_rot);
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemsetAsync(b, 1, bytes, s)); }
        for (int grid : {512, 1024, 2048, 4096}) {
            for (int cold = 0; cold < 2; ++cold) {
                float best[2] = {0, 0};
                for (int variant = 0; variant < 2; ++variant) {
                    const int iters = 200;
                    for (int rep = 0; rep < 2; ++rep) {
                        CK(hipEventRecord(e0, s));
                        for (int i = 0; i < iters; ++i) {
                            const float4* p = bufs[cold ? i % n_rot : 0];
                            if (variant == 0) hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, s, p, n4, out);
                            else hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, s, p, n4, out);
                        }
                        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        best[variant] = ms * 1e3f / iters;
                    }
                }
                printf("%6.1f MB  grid %4d  %s : U=4 %6.2f us = %5.2f TB/s | U=8 %6.2f us = %5.2f TB/s\n", mb, grid,
                       cold ? "HBM (rotating)" : "same buffer   ", best[0], bytes / best[0] * 1e-6, best[1], bytes / best[1] * 1e-6);
            }
        }
        for (auto& b : bufs) CK(hipFree(b));
    }
    return 0;
}
