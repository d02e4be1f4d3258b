// Embedding lookup / dense gradient for the GloVe table shared by the text encoder and the
// decoder (train.py:25-31, model/encoder.py:96, model/decoder.py:75).  Rows are 4*E bytes
// (1200 B at E=300): one wavefront copies one row with 16-byte lanes, four rows per
// workgroup.  The gradient is the dense scatter-add torch produces for sparse=False:
// f32 atomics, one contiguous row segment per wave instruction (the shape the memory-side
// atomic units run at full rate for).
#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

__global__ __launch_bounds__(256) void embedding_fwd_kernel(const float* __restrict__ table,
                                                            const int64_t* __restrict__ ids, float* __restrict__ out,
                                                            int n, int V, int E, int ld_out, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n) return;
    const int64_t id = ids[r];
    const bool ok = id >= 0 && id < V;
    const float* src = table + (ok ? id : 0) * (int64_t)E;
    float* dst = out + (int64_t)r * ld_out;
    if (vec) {
        for (int c = 4 * lane; c < E; c += 256) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) x = *reinterpret_cast<const float4*>(src + c);
            *reinterpret_cast<float4*>(dst + c) = x;
        }
    } else {
        for (int c = lane; c < E; c += 64) dst[c] = ok ? src[c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dout, int ld,
                                                            const int64_t* __restrict__ ids, float* __restrict__ dtable,
                                                            int n, int V, int E) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n) return;
    const int64_t id = ids[r];
    if (id < 0 || id >= V) return;
    const float* src = dout + (int64_t)r * ld;
    float* dst = dtable + id * (int64_t)E;
    for (int c = lane; c < E; c += 64) atomicAdd(dst + c, src[c]);
}

}  // namespace

namespace mmqg {

int embedding_fwd(const float* table, const int64_t* ids, float* out, int n, int V, int E, int ld_out, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && V > 0 && E > 0 && ld_out >= E, "embedding_fwd: bad shape");
    if (n == 0) return 0;
    MMQG_REQUIRE(table && ids && out, "embedding_fwd: null pointer");
    const int vec = (E % 4 == 0) && (ld_out % 4 == 0) && aligned16(table) && aligned16(out);
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, table, ids, out, n, V, E, ld_out, vec);
    return check_launch("embedding_fwd");
}

int embedding_bwd(const float* dout, int ld, const int64_t* ids, float* dtable, int n, int V, int E, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && V > 0 && E > 0 && ld >= E, "embedding_bwd: bad shape");
    if (n == 0) return 0;
    MMQG_REQUIRE(dout && ids && dtable, "embedding_bwd: null pointer");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, dout, ld, ids, dtable, n, V, E);
    return check_launch("embedding_bwd");
}

}  // namespace mmqg
