// Cross entropy over the vocabulary logits with argmax (train.py:174,264; the greedy pick of
// train.py:107-108) plus the small reductions the gradient path needs (bias gradients =
// column sums, the scalar loss).  One workgroup per logits row: three sweeps over the row
// (max+argmax, sum of exp, gradient write); a row is 40 KB at V=10k so sweeps two and three
// are L2 hits.
#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

// stats (nullable): [rows][stats_tiles] partial {max, sum exp(x - max), argmax index bits} per column tile, written by the
// projection GEMM's epilogue (gemm_nt_tile.hip): sweeps 1 and 2 then shrink to combining stats_tiles entries
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, int ld,
                                                         const int64_t* __restrict__ target,
                                                         const float* __restrict__ row_weight, int V,
                                                         float* __restrict__ loss_rows, int64_t* __restrict__ argmax,
                                                         float* dlogits, int ld_d, const float4* __restrict__ stats,
                                                         int stats_tiles) {
    __shared__ float shv[4];
    __shared__ int shi[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (int64_t)r * ld;
    // sweep 1: max and the FIRST index that attains it (torch.argmax tie rule)
    float best = -INFINITY; int bi = 0x7fffffff;
    const float4* st = stats ? stats + (int64_t)r * stats_tiles : nullptr;
    if (st) {
        for (int c = tid; c < stats_tiles; c += 256) {
            const float4 t = st[c];
            const int ti = __float_as_int(t.z);
            if (t.x > best || (t.x == best && ti < bi)) { best = t.x; bi = ti; }
        }
    } else {
        for (int c = tid; c < V; c += 256) {
            const float x = row[c];
            if (x > best || (x == best && c < bi)) { best = x; bi = c; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { shv[wave] = best; shi[wave] = bi; }
    __syncthreads();
    best = shv[0]; bi = shi[0];
    for (int w = 1; w < 4; ++w)
        if (shv[w] > best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
    __syncthreads();
    // sweep 2: sum exp
    float part = 0.f;
    if (st) {
        for (int c = tid; c < stats_tiles; c += 256) { const float4 t = st[c]; part += t.y * expf(t.x - best); }
    } else {
        for (int c = tid; c < V; c += 256) part += expf(row[c] - best);
    }
    part = wave_sum(part);
    if (lane == 0) shv[wave] = part;
    __syncthreads();
    const float sum = shv[0] + shv[1] + shv[2] + shv[3];
    const float lse = best + logf(sum);
    const float wgt = row_weight ? row_weight[r] : 1.f;
    int64_t tg = target ? target[r] : -1;
    const bool tg_ok = tg >= 0 && tg < V;
    if (tid == 0) {
        if (argmax) argmax[r] = bi;
        if (loss_rows) loss_rows[r] = (tg_ok && wgt != 0.f) ? wgt * (lse - row[tg]) : 0.f;
    }
    // sweep 3: gradient (may overwrite the logits row in place: every thread has finished
    // reading other columns only after this barrier)
    if (dlogits) {
        __syncthreads();
        float* drow = dlogits + (int64_t)r * ld_d;
        const float inv = 1.0f / sum;
        for (int c = tid; c < V; c += 256) {
            float g = 0.f;
            if (tg_ok && wgt != 0.f) g = wgt * (expf(row[c] - best) * inv - (c == tg ? 1.f : 0.f));
            drow[c] = g;
        }
    }
}

// Gumbel-max sampling: argmax_c (logits[c] - log(-log(u_c))) is a draw from softmax(logits)
__global__ __launch_bounds__(256) void sample_gumbel_kernel(const float* __restrict__ logits, int ld, int V,
                                                            uint64_t seed, uint64_t stream_id,
                                                            int64_t* __restrict__ out_ids) {
    __shared__ float shv[4];
    __shared__ int shi[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (int64_t)r * ld;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = tid; c < V; c += 256) {
        uint32_t rnd[4];
        const uint64_t idx = (uint64_t)r * (uint64_t)V + c;
        philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream_id,
                      (uint32_t)(stream_id >> 32), rnd);
        const float u = ((float)(rnd[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);     // (0,1)
        const float x = row[c] - logf(-logf(u));
        if (x > best || (x == best && c < bi)) { best = x; bi = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { shv[wave] = best; shi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (shv[w] > best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
        out_ids[r] = bi;
    }
}

// out[n] += sum_m X[m][n]; rows split over blockIdx.y, finished with f32 atomics
__global__ __launch_bounds__(256) void colsum_add_kernel(const float* __restrict__ X, int ld, int M, int N,
                                                         float* __restrict__ out, float* __restrict__ out2,
                                                         int rows_per_block) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float acc = 0.f;
    for (int m = m0; m < m1; ++m) acc += X[(int64_t)m * ld + n];
    atomicAdd(out + n, acc);
    if (out2) atomicAdd(out2 + n, acc);
}

// aligned case: a workgroup covers 256 columns x rows_per_block rows as 64 float4 column lanes x 4 row lanes,
// the row lanes are combined through LDS, one atomic per column and output
__global__ __launch_bounds__(256) void colsum_add_vec4_kernel(const float* __restrict__ X, int ld, int M, int N,
                                                              float* __restrict__ out, float* __restrict__ out2,
                                                              int rows_per_block) {
    __shared__ float4 part[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + 4 * cl;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
        for (int m = m0 + rl; m < m1; m += 4) {
            const float4 v = *reinterpret_cast<const float4*>(X + (int64_t)m * ld + n);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && n < N) {
        const float4 a = part[0][cl], b = part[1][cl], c = part[2][cl], d = part[3][cl];
        const float s0 = a.x + b.x + c.x + d.x, s1 = a.y + b.y + c.y + d.y, s2 = a.z + b.z + c.z + d.z, s3 = a.w + b.w + c.w + d.w;
        atomicAdd(out + n, s0); atomicAdd(out + n + 1, s1); atomicAdd(out + n + 2, s2); atomicAdd(out + n + 3, s3);
        if (out2) { atomicAdd(out2 + n, s0); atomicAdd(out2 + n + 1, s1); atomicAdd(out2 + n + 2, s2); atomicAdd(out2 + n + 3, s3); }
    }
}

__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float sh[4];
    float part = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) part += x[i];
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha,
                                                   int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void fill_copy_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                        int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src ? src[i] : 0.f;
}

// several independent fills / copies in one launch (initial states of a layer stack): blockIdx.y = segment
__global__ __launch_bounds__(256) void fill_copy_multi_kernel(mmqg::CopyBatch b) {
    const mmqg::CopySeg& c = b.seg[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < c.n; i += stride) c.dst[i] = c.src ? c.src[i] : 0.f;
}

__global__ __launch_bounds__(256) void fill_i64_kernel(int64_t* __restrict__ dst, int64_t value, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = value;
}

__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ dst, int64_t dst_stride,
                                                       const float* __restrict__ src, int64_t src_stride, int rows,
                                                       int cols) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[r * dst_stride + c] += src[r * src_stride + c];
}

}  // namespace

namespace mmqg {

int ce_fwd_bwd(const float* logits, int ld, const int64_t* target, const float* row_weight, int rows, int V,
               float* loss_rows, int64_t* argmax, float* dlogits, int ld_d, hipStream_t s, const float* stats,
               int stats_tiles) {
    MMQG_REQUIRE(rows >= 0 && V > 0 && ld >= V, "ce_fwd_bwd: bad shape");
    if (rows == 0) return 0;
    MMQG_REQUIRE(logits, "ce_fwd_bwd: null logits");
    MMQG_REQUIRE(!dlogits || ld_d >= V, "ce_fwd_bwd: ld_d < V");
    MMQG_REQUIRE(target || (!loss_rows && !dlogits), "ce_fwd_bwd: loss/gradient requested without targets");
    const float4* st = (stats && stats_tiles > 0) ? reinterpret_cast<const float4*>(stats) : nullptr;
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(rows), dim3(256), 0, s, logits, ld, target, row_weight, V, loss_rows,
                       argmax, dlogits, ld_d, st, stats_tiles);
    return check_launch("ce_fwd_bwd");
}

int sample_gumbel(const float* logits, int ld, int rows, int V, uint64_t seed, uint64_t stream_id, int64_t* out_ids,
                  hipStream_t s) {
    MMQG_REQUIRE(rows >= 0 && V > 0 && ld >= V && logits && out_ids, "sample_gumbel: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(sample_gumbel_kernel, dim3(rows), dim3(256), 0, s, logits, ld, V, seed, stream_id, out_ids);
    return check_launch("sample_gumbel");
}

int colsum_add(const float* X, int ld, int M, int N, float* out, hipStream_t s) {
    return colsum_add2(X, ld, M, N, out, nullptr, s);
}

int colsum_add2(const float* X, int ld, int M, int N, float* out, float* out2, hipStream_t s) {
    MMQG_REQUIRE(M >= 0 && N >= 0 && ld >= N, "colsum_add: bad shape");
    if (M == 0 || N == 0) return 0;
    MMQG_REQUIRE(X && out, "colsum_add: null pointer");
    if (N % 4 == 0 && ld % 4 == 0 && aligned16(X)) {
        // ~16 rows per workgroup: enough workgroups to fill the chip on the [T*B][4H] gradients
        const int rows_per_block = std::max(16, ceil_div(M, 512));
        hipLaunchKernelGGL(colsum_add_vec4_kernel, dim3(ceil_div(N, 256), ceil_div(M, rows_per_block)), dim3(256), 0, s, X, ld, M,
                           N, out, out2, rows_per_block);
        return check_launch("colsum_add");
    }
    int slices = std::min(64, std::max(1, M / 32));
    const int rows_per_block = ceil_div(M, slices);
    slices = ceil_div(M, rows_per_block);
    hipLaunchKernelGGL(colsum_add_kernel, dim3(ceil_div(N, 256), slices), dim3(256), 0, s, X, ld, M, N, out, out2,
                       rows_per_block);
    return check_launch("colsum_add");
}

int reduce_sum(const float* x, int n, float* out, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && x && out, "reduce_sum: bad arguments");
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s, x, n, out);
    return check_launch("reduce_sum");
}

int axpy(float* y, const float* x, float alpha, int64_t n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && x && y, "axpy: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, y, x, alpha, n);
    return check_launch("axpy");
}

// dst[0:n] = src ? src[0:n] : 0.  Own kernel instead of hipMemsetAsync / hipMemcpyAsync: memset nodes
// captured into a hipGraph were observed (ROCm 7.2, gfx950) to race with neighbouring kernel nodes.
int copy_or_zero_f32(float* dst, const float* src, int64_t n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && dst, "copy_or_zero_f32: bad arguments");
    if (n == 0) return 0;
    const int64_t blocks = std::min<int64_t>(ceil_div64(n, 256), 2048);
    hipLaunchKernelGGL(fill_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, src, n);
    return check_launch("copy_or_zero_f32");
}

int copy_or_zero_multi(const CopySeg* segs, int n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && (n == 0 || segs), "copy_or_zero_multi: bad arguments");
    for (int i0 = 0; i0 < n; i0 += kMaxCopySegs) {
        CopyBatch b{};
        const int m = std::min(kMaxCopySegs, n - i0);
        int64_t longest = 0;
        for (int i = 0; i < m; ++i) {
            MMQG_REQUIRE(segs[i0 + i].n >= 0 && (segs[i0 + i].n == 0 || segs[i0 + i].dst), "copy_or_zero_multi: bad segment");
            b.seg[i] = segs[i0 + i];
            longest = std::max(longest, segs[i0 + i].n);
        }
        if (longest == 0) continue;
        const int64_t blocks = std::min<int64_t>(ceil_div64(longest, 256), 1024);
        hipLaunchKernelGGL(fill_copy_multi_kernel, dim3((unsigned)blocks, m), dim3(256), 0, s, b);
        MMQG_TRY(check_launch("copy_or_zero_multi"));
    }
    return 0;
}

int fill_i64(int64_t* dst, int64_t value, int64_t n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && dst, "fill_i64: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, dst, value, n);
    return check_launch("fill_i64");
}

int add_rows_strided(float* dst, int64_t dst_stride, const float* src, int64_t src_stride, int rows, int cols,
                     hipStream_t s) {
    MMQG_REQUIRE(rows >= 0 && cols >= 0 && dst && src, "add_rows_strided: bad arguments");
    if (rows == 0 || cols == 0) return 0;
    const int64_t n = (int64_t)rows * cols;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, dst, dst_stride, src,
                       src_stride, rows, cols);
    return check_launch("add_rows_strided");
}

}  // namespace mmqg
