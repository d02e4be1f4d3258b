// Probe for a persistent time-loop design on MI355X: what does a device-wide barrier cost next to a
// kernel boundary, and how fast can every workgroup re-read a block of activations that all workgroups
// just wrote (cross-XCD visibility checked)?
//   hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.hip -o tools/barrier_probe && tools/barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Bar { unsigned count; unsigned gen; unsigned fail; unsigned pad; };

// sense-reversal barrier over all workgroups of the grid; bounded spin so a scheduling surprise ends in an
// error flag instead of a hang
__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nblocks, unsigned& my_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();   // release: this workgroup's global writes
        const unsigned target = my_gen + 1;
        if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || __hip_atomic_load(&b->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&b->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        __threadfence();   // acquire
    }
    my_gen += 1;
    ok = __syncthreads_and(ok);
    return ok;
}

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 1000000) p[0] = 1.f; }

__global__ __launch_bounds__(512) void barrier_loop(Bar* b, int iters) {
    unsigned gen = 0;
    for (int i = 0; i < iters; ++i)
        if (!grid_barrier(b, gridDim.x, gen)) return;
}

// every iteration: workgroup w writes its slice of act[(i+1)&1] (values depend on i), barrier, then EVERY
// workgroup reads all `n_act` floats of that buffer and checks the sum
__global__ __launch_bounds__(512) void barrier_data_loop(Bar* b, int iters, float* act, int n_act, unsigned* errors, int read_all) {
    unsigned gen = 0;
    const int per = n_act / gridDim.x;
    __shared__ float red[8];
    for (int i = 0; i < iters; ++i) {
        float* dst = act + (size_t)(i & 1) * n_act;
        for (int k = threadIdx.x; k < per; k += blockDim.x) dst[blockIdx.x * per + k] = (float)(i + 1);
        if (!grid_barrier(b, gridDim.x, gen)) return;
        if (read_all) {
            float s = 0.f;
            const float4* src = reinterpret_cast<const float4*>(dst);
            for (int k = threadIdx.x; k < n_act / 4; k += blockDim.x) {
                const float4 v = src[k];
                s += v.x + v.y + v.z + v.w;
            }
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
                for (int w = 0; w < blockDim.x / 64; ++w) t += red[w];
                if (t != (float)(i + 1) * n_act) atomicAdd(errors, 1u);
            }
            __syncthreads();
        }
    }
}

int main(int argc, char** argv) {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Bar* bar; CK(hipMalloc(&bar, sizeof(Bar)));
    unsigned* errors; CK(hipMalloc(&errors, 4));
    const int iters = 200;
    float ms;

    // 1. chain of dependent launches, eager and as a graph
    for (int grid : {256, 512}) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(512), 0, s, nullptr);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(512), 0, s, nullptr);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager chain  grid %4d: %.2f us per launch\n", grid, ms * 1e3 / iters);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(512), 0, s, nullptr);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph chain  grid %4d: %.2f us per launch\n", grid, ms * 1e3 / iters);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }

    // 2. barrier only
    for (int grid : {128, 256, 512}) {
        for (int threads : {256, 512}) {
            CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
            hipLaunchKernelGGL(barrier_loop, dim3(grid), dim3(threads), 0, s, bar, 10);
            CK(hipStreamSynchronize(s));
            CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(barrier_loop, dim3(grid), dim3(threads), 0, s, bar, iters);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            Bar hb; CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
            printf("barrier      grid %4d x %3d: %.2f us per barrier%s\n", grid, threads, ms * 1e3 / iters, hb.fail ? "  [SPIN TIMEOUT]" : "");
            if (hb.fail) return 2;
        }
    }

    // 3. write slice -> barrier -> everybody reads everything (activation exchange of a layer-step)
    for (int kb : {128, 256, 768}) {
        const int n_act = kb * 1024 / 4;
        float* act; CK(hipMalloc(&act, (size_t)2 * n_act * 4));
        for (int grid : {256, 512}) {
            for (int read_all : {0, 1}) {
                CK(hipMemsetAsync(bar, 0, sizeof(Bar), s)); CK(hipMemsetAsync(errors, 0, 4, s));
                hipLaunchKernelGGL(barrier_data_loop, dim3(grid), dim3(512), 0, s, bar, 10, act, n_act, errors, read_all);
                CK(hipStreamSynchronize(s));
                CK(hipMemsetAsync(bar, 0, sizeof(Bar), s)); CK(hipMemsetAsync(errors, 0, 4, s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(barrier_data_loop, dim3(grid), dim3(512), 0, s, bar, iters, act, n_act, errors, read_all);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                Bar hb; unsigned he;
                CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, errors, 4, hipMemcpyDeviceToHost));
                printf("exchange %3d KB grid %4d read_all %d: %.2f us per step, stale reads %u%s\n", kb, grid, read_all,
                       ms * 1e3 / iters, he, hb.fail ? "  [SPIN TIMEOUT]" : "");
                if (hb.fail) return 2;
            }
        }
        CK(hipFree(act));
    }
    printf("done\n");
    return 0;
}
