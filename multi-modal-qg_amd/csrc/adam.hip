// torch.optim.Adam (default betas/eps, no weight decay, no amsgrad) over a flat fp32 range,
// as stepped three times per question at train.py:179-181.  One fused pass: 16 bytes read
// (p,g,m,v) + 12 written per parameter -> HBM-bound.  The 1-based step number lives in device
// memory so the launch can sit inside a replayed hipGraph; the bias corrections are formed
// in double precision like the host-side Python of torch.optim does.
#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, double lr,
                                                   double b1d, double b2d, float eps, const int32_t* __restrict__ step,
                                                   float grad_scale, float omb1, float omb2,
                                                   const int32_t* __restrict__ skip) {
    __shared__ float sh[3];
    const float b1 = (float)b1d, b2 = (float)b2d;
    if (threadIdx.x == 0) {
        const double t = (double)step[0];
        const double bc1 = 1.0 - pow(b1d, t);
        const double bc2 = 1.0 - pow(b2d, t);
        sh[0] = (float)(lr / bc1);     // step size
        sh[1] = (float)sqrt(bc2);              // sqrt of the second-moment correction
        // skip[0] != 0 (mmqg_persist_guard_refresh): a persistent time loop of this step timed out at its barrier
        // (persist_rt.hip) and its gradients are NaN-poisoned.  Dropping the update keeps the parameters and both
        // moments intact — the host raises at its next health check and the run can go on after
        // mmqg_persist_clear_failures() instead of needing a checkpoint restore.
        sh[2] = (skip && skip[0] != 0) ? 1.f : 0.f;
    }
    __syncthreads();
    if (sh[2] != 0.f) return;
    const float step_size = sh[0], bc2_sqrt = sh[1];
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 pp = *reinterpret_cast<float4*>(p + i);
            float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gr = ga[k] * grad_scale;
                ma[k] = b1 * ma[k] + omb1 * gr;
                va[k] = b2 * va[k] + omb2 * gr * gr;
                pa[k] -= step_size * ma[k] / (sqrtf(va[k]) / bc2_sqrt + eps);
            }
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (int64_t j = i; j < n; ++j) {
                const float gr = g[j] * grad_scale;
                const float mj = b1 * m[j] + omb1 * gr;
                const float vj = b2 * v[j] + omb2 * gr * gr;
                m[j] = mj; v[j] = vj;
                p[j] -= step_size * mj / (sqrtf(vj) / bc2_sqrt + eps);
            }
        }
    }
}

__global__ void counter_add_kernel(int32_t* ctr, int delta) { ctr[0] += delta; }

// flag[0] = has any persistent launch of this process reported a failure (the word in pinned host memory; ONE lane
// reads it: 2,048 workgroups of the Adam kernel doing so took 0.6 ms per launch)
__global__ void guard_refresh_kernel(int32_t* flag, const unsigned* host) {
    flag[0] = (host && __hip_atomic_load(host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) ? 1 : 0;
}

}  // namespace

namespace mmqg {

int adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps,
              const int32_t* step, float grad_scale, hipStream_t s, const int32_t* skip) {
    MMQG_REQUIRE(n >= 0, "adam_step: negative length");
    if (n == 0) return 0;
    MMQG_REQUIRE(p && g && m && v && step, "adam_step: null pointer");
    MMQG_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adam_step: buffers must be 16-byte aligned");
    const int64_t blocks = std::min<int64_t>(ceil_div64(n, 1024), 2048);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2,
                       (float)eps, step, grad_scale, (float)(1.0 - b1), (float)(1.0 - b2),
                       skip);
    return check_launch("adam_step");
}

int persist_guard_refresh(int32_t* flag, hipStream_t s) {
    MMQG_REQUIRE(flag, "persist_guard_refresh: null pointer");
    persist_runtime_prepare();
    hipLaunchKernelGGL(guard_refresh_kernel, dim3(1), dim3(1), 0, s, flag, (const unsigned*)persist_host_fail_word());
    return check_launch("persist_guard_refresh");
}

int counter_add(int32_t* ctr, int delta, hipStream_t s) {
    MMQG_REQUIRE(ctr, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, s, ctr, delta);
    return check_launch("counter_add");
}

}  // namespace mmqg
