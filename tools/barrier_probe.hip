// Probe for a persistent time-loop design on MI355X (VERDICT r1 #4 redo): what does a device-wide barrier
// cost next to a kernel boundary when it is built the way MI355X_MICROARCH.md's price list builds it?
//
//   flat     one monotonic counter on a 128-byte line of its own, relaxed sc1-load polls with s_sleep,
//            lane-0 release fence before the arrival, one acquire fence after the wait   (row barrier-counter)
//   xcd      per-XCC counter -> the XCC's last arriver (leader) does the release fence, adds to the top
//            counter, waits for all XCCs, acquires, then bumps its XCC's generation word; every other
//            workgroup polls its XCC's generation word and acquires                        (row barrier-xcd)
//   *_nofence the same without any fence: legal when every exchanged byte is stored sc1 (write-through),
//            drained with s_waitcnt vmcnt(0) before the arrival, and loaded with sc1 loads (Guideline 16, R1)
//
// and the exchange a recurrent layer-step needs: every workgroup writes its slice of an activation block,
// barrier, every workgroup reads the WHOLE block (checked word for word against the iteration's values,
// with the consumer's L1 warm: the block of the previous iteration was read from the same addresses).
//
//   hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.hip -o tools/barrier_probe && tools/barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
constexpr int LINE = 32;   // u32 per 128-byte line

struct XBar {
    u32 xcc_count[8 * LINE];
    u32 top[LINE];
    u32 xcc_gen[8 * LINE];
    u32 flat[LINE];
    u32 census[8 * LINE];
    u32 fail[LINE];
};

__device__ __forceinline__ u32 xcc_id() {
    u32 v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ u32 ld_rlx(u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 add_rlx(u32* p, u32 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bounded spin: a scheduling surprise ends in an error word, never in a hang
__device__ __forceinline__ bool spin_until_ge(u32* p, u32 target, u32* fail) {
    u32 spins = 0;
    while ((int)(ld_rlx(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 21) || ((spins & 255u) == 0 && ld_rlx(fail))) { st_rlx(fail, 1u); return false; }
    }
    return true;
}
__device__ __forceinline__ void release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ROCm 7.2 may drop the fence's own wait (Guideline 16 pitfall 12)
}
__device__ __forceinline__ void acquire_agent() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FENCE>
__device__ __forceinline__ bool bar_flat(XBar* b, u32 n, u32 k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its stores
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (FENCE) release_agent();
        add_rlx(&b->flat[0], 1u);
        ok = spin_until_ge(&b->flat[0], k * n, b->fail);
        if (FENCE) acquire_agent();
    }
    return __syncthreads_and(ok);
}

template <bool FENCE>
__device__ __forceinline__ bool bar_xcd(XBar* b, u32 x, u32 n_x, u32 n_groups, u32 k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const u32 v = add_rlx(&b->xcc_count[x * LINE], 1u);
        if (v == k * n_x - 1) {                          // last arriver of this XCC: its L2 now holds every slice of the XCC
            if (FENCE) release_agent();
            add_rlx(&b->top[0], 1u);
            ok = spin_until_ge(&b->top[0], k * n_groups, b->fail);
            if (FENCE) acquire_agent();
            st_rlx(&b->xcc_gen[x * LINE], k);
        } else {
            ok = spin_until_ge(&b->xcc_gen[x * LINE], k, b->fail);
            if (FENCE) acquire_agent();
        }
    }
    return __syncthreads_and(ok);
}

// census of workgroups per XCC + one flat barrier; returns n_x (mine) and n_groups
__device__ __forceinline__ bool census(XBar* b, u32& x, u32& n_x, u32& n_groups) {
    x = xcc_id();
    if (threadIdx.x == 0) add_rlx(&b->census[x * LINE], 1u);
    if (!bar_flat<true>(b, gridDim.x, 1)) return false;
    n_x = ld_rlx(&b->census[x * LINE]);
    n_groups = 0;
    for (int i = 0; i < 8; ++i) n_groups += ld_rlx(&b->census[i * LINE]) ? 1u : 0u;
    return true;
}

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 1000000) p[0] = 1.f; }

template <int MODE>   // 0 flat+fence, 1 flat nofence, 2 xcd+fence, 3 xcd nofence
__global__ __launch_bounds__(512) void barrier_loop(XBar* b, int iters, u32* info) {
    u32 x, n_x, n_g;
    if (!census(b, x, n_x, n_g)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && info) { info[0] = n_g; for (int i = 0; i < 8; ++i) info[1 + i] = ld_rlx(&b->census[i * LINE]); }
    for (int i = 0; i < iters; ++i) {
        bool ok;
        if (MODE == 0) ok = bar_flat<true>(b, gridDim.x, 2 + i);
        else if (MODE == 1) ok = bar_flat<false>(b, gridDim.x, 2 + i);
        else if (MODE == 2) ok = bar_xcd<true>(b, x, n_x, n_g, 1 + i);
        else ok = bar_xcd<false>(b, x, n_x, n_g, 1 + i);
        if (!ok) return;
    }
}

__device__ __forceinline__ u32 expect(int i, int idx) { return (u32)(i * 2654435761u) ^ (u32)(idx * 40503u + 17u); }

// every iteration: workgroup w writes its slice of act[i&1], barrier, EVERY workgroup reads the whole block and
// checks every word.  SC1 = write-through stores + sc1 loads with the fence-free barrier; else plain + fences.
template <bool XCD, bool SC1>
__global__ __launch_bounds__(512) void exchange_loop(XBar* b, int iters, u32* act, int n_words, u32* errors, int read_all) {
    u32 x, n_x, n_g;
    if (!census(b, x, n_x, n_g)) return;
    const int per = n_words / gridDim.x;            // words per workgroup slice (multiple of 4)
    u32 bad = 0;
    for (int i = 0; i < iters; ++i) {
        u32* dst = act + (size_t)(i & 1) * n_words;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, n_words * 4, 0x00020000);
        for (int k = threadIdx.x * 4; k < per; k += blockDim.x * 4) {
            const int idx = blockIdx.x * per + k;
            u32x4 v = {expect(i, idx), expect(i, idx + 1), expect(i, idx + 2), expect(i, idx + 3)};
            if (SC1) __builtin_amdgcn_raw_buffer_store_b128(v, rs, idx * 4, 0, 16);
            else *reinterpret_cast<u32x4*>(dst + idx) = v;
        }
        bool ok;
        if (XCD) ok = SC1 ? bar_xcd<false>(b, x, n_x, n_g, 1 + i) : bar_xcd<true>(b, x, n_x, n_g, 1 + i);
        else ok = SC1 ? bar_flat<false>(b, gridDim.x, 2 + i) : bar_flat<true>(b, gridDim.x, 2 + i);
        if (!ok) return;
        if (read_all) {
            // 8 independent 16-byte loads in flight per lane (a latency-bound read loop would hide the real rate)
            const int stride = blockDim.x * 4;
            for (int k0 = threadIdx.x * 4; k0 < n_words; k0 += 8 * stride) {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * stride;
                    if (k < n_words) {
                        if (SC1) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, k * 4, 0, 16);
                        else v[u] = *reinterpret_cast<const u32x4*>(dst + k);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * stride;
                    if (k < n_words)
                        bad += (v[u].x != expect(i, k)) + (v[u].y != expect(i, k + 1)) + (v[u].z != expect(i, k + 2)) + (v[u].w != expect(i, k + 3));
                }
            }
        }
    }
    if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    XBar* bar; CK(hipMalloc(&bar, sizeof(XBar)));
    u32* errors; CK(hipMalloc(&errors, 64));
    u32* info; CK(hipMalloc(&info, 64));
    const int iters = 400;
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

    // 2. barrier only (host-paired wall time of `iters` barriers; the launch itself and the census are inside: ~10 us / iters)
    const char* names[4] = {"flat  + fences ", "flat  no fence ", "xcd   + fences ", "xcd   no fence "};
    for (int grid : {256, 512}) {
        for (int threads : {256, 512}) {
            if (grid * threads > 256 * 512) continue;            // stay at <= 1 resident 512-thread workgroup per CU
            for (int mode = 0; mode < 4; ++mode) {
                float best = 1e30f;
                bool failed = false;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(bar, 0, sizeof(XBar), s));
                    CK(hipEventRecord(e0, s));
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(barrier_loop<0>, dim3(grid), dim3(threads), 0, s, bar, iters, info); break;
                        case 1: hipLaunchKernelGGL(barrier_loop<1>, dim3(grid), dim3(threads), 0, s, bar, iters, info); break;
                        case 2: hipLaunchKernelGGL(barrier_loop<2>, dim3(grid), dim3(threads), 0, s, bar, iters, info); break;
                        default: hipLaunchKernelGGL(barrier_loop<3>, dim3(grid), dim3(threads), 0, s, bar, iters, info); break;
                    }
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                    XBar hb; CK(hipMemcpy(&hb, bar, sizeof(XBar), hipMemcpyDeviceToHost));
                    if (hb.fail[0]) failed = true;
                    if (ms < best) best = ms;
                }
                u32 hi[9]; CK(hipMemcpy(hi, info, 36, hipMemcpyDeviceToHost));
                printf("barrier %s grid %4d x %3d: %6.2f us per barrier%s", names[mode], grid, threads, best * 1e3 / iters, failed ? "  [SPIN TIMEOUT]" : "");
                if (mode == 2) { printf("   (XCC groups %u:", hi[0]); for (int i = 0; i < 8; ++i) printf(" %u", hi[1 + i]); printf(")"); }
                printf("\n");
                if (failed) return 2;
            }
        }
    }

    // 3. write slice -> barrier -> everybody reads everything (the activation exchange of a recurrent layer-step)
    for (int kb : {128, 256, 512}) {
        const int n_words = kb * 1024 / 4;
        u32* act; CK(hipMalloc(&act, (size_t)2 * n_words * 4));
        for (int variant = 0; variant < 4; ++variant) {
            for (int read_all : {0, 1}) {
                float best = 1e30f; u32 he = 0; bool failed = false;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(bar, 0, sizeof(XBar), s)); CK(hipMemsetAsync(errors, 0, 4, s));
                    CK(hipMemsetAsync(act, 0, (size_t)2 * n_words * 4, s));
                    CK(hipEventRecord(e0, s));
                    switch (variant) {
                        case 0: hipLaunchKernelGGL((exchange_loop<false, false>), dim3(256), dim3(512), 0, s, bar, iters, act, n_words, errors, read_all); break;
                        case 1: hipLaunchKernelGGL((exchange_loop<false, true>), dim3(256), dim3(512), 0, s, bar, iters, act, n_words, errors, read_all); break;
                        case 2: hipLaunchKernelGGL((exchange_loop<true, false>), dim3(256), dim3(512), 0, s, bar, iters, act, n_words, errors, read_all); break;
                        default: hipLaunchKernelGGL((exchange_loop<true, true>), dim3(256), dim3(512), 0, s, bar, iters, act, n_words, errors, read_all); break;
                    }
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                    XBar hb; u32 e;
                    CK(hipMemcpy(&hb, bar, sizeof(XBar), hipMemcpyDeviceToHost)); CK(hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost));
                    he += e; if (hb.fail[0]) failed = true;
                    if (ms < best) best = ms;
                }
                const char* vn[4] = {"flat, plain + fences", "flat, sc1 no fence  ", "xcd,  plain + fences", "xcd,  sc1 no fence  "};
                printf("exchange %3d KB  %s  read_all %d: %6.2f us per step, wrong words %u%s\n", kb, vn[variant], read_all,
                       best * 1e3 / iters, he, failed ? "  [SPIN TIMEOUT]" : "");
                if (failed) return 2;
            }
        }
        CK(hipFree(act));
    }
    printf("done\n");
    return 0;
}
