// Frame CNN of VideoConvLstmEncoder (model/encoder.py:40-50,64-67): four blocks of
//   3x3 valid convolution (stride 1) -> ReLU -> BatchNorm2d, with a 3x3/3 max-pool after blocks 2 and 4.
// Channel counts are tiny (3 -> 4 -> 6 -> 8 -> 10), so this is memory-bound element-wise work, not a
// GEMM: direct convolution, one thread per output pixel computing every output channel, weights in
// LDS.  BatchNorm statistics are per QUESTION (the reference feeds one question's T frames as the
// batch, encoder.py:64), gathered by the convolution kernel itself (block reduction + f64 atomics),
// so a block costs: conv+ReLU+stats, a tiny finalize, normalise(+pool).  Frames past a question's
// n_frames are excluded from the statistics and produce zero features.
//
// Backward mirrors it: pooled gradients are routed to the window arg-max (recorded by the forward),
// the two BatchNorm reductions run over the pooled domain, then one pass forms the gradient of the
// convolution output (BN + ReLU), and two kernels give the input and the weight/bias gradients.
#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

constexpr int kMaxC = 16;      // channel bound of the per-thread accumulators

// frame index -> (question, time): frames are stored [B][T] (the drop-in modules) or [T][B] (the batched
// trainer, whose frame LSTM is time-major)
__device__ __forceinline__ void frame_bt(int n, int N, int T, int tm, int& b, int& t) {
    if (tm) { const int B = N / T; b = n % B; t = n / B; } else { b = n / T; t = n % T; }
}

struct ConvK {
    const float* x;            // [N][Cin][H][W]
    const float* w;            // [Cout][Cin][3][3]
    const float* bias;         // [Cout]
    float* y;                  // [N][Cout][H-2][W-2]  relu(conv)
    double* stats;             // [B][Cout][2] sum, sum of squares over the question's valid frames (nullable)
    const int32_t* n_frames;   // [B] nullable = all T valid
    int N, T, Cin, Cout, H, W, tm;
};

__global__ __launch_bounds__(256) void conv3x3_relu_stats_kernel(ConvK a) {
    __shared__ float ws[kMaxC * kMaxC * 9];
    __shared__ float red[4][2 * kMaxC];
    const int n = blockIdx.y, Ho = a.H - 2, Wo = a.W - 2;
    for (int i = threadIdx.x; i < a.Cout * a.Cin * 9; i += 256) ws[i] = a.w[i];
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const bool in = pix < Ho * Wo;
    const int oy = in ? pix / Wo : 0, ox = in ? pix % Wo : 0;
    float acc[kMaxC];
#pragma unroll
    for (int co = 0; co < kMaxC; ++co) acc[co] = co < a.Cout ? a.bias[co] : 0.f;
    if (in) {
        const float* xp = a.x + (int64_t)n * a.Cin * a.H * a.W + (int64_t)oy * a.W + ox;
        for (int ci = 0; ci < a.Cin; ++ci) {
            float v[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = xp[(int64_t)ci * a.H * a.W + dy * a.W + dx];
#pragma unroll
            for (int co = 0; co < kMaxC; ++co) {
                if (co < a.Cout) {
                    const float* wr = ws + (co * a.Cin + ci) * 9;
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc[co] += v[k] * wr[k];
                }
            }
        }
    }
    int b, t; frame_bt(n, a.N, a.T, a.tm, b, t);
    const bool valid = a.n_frames ? (t < a.n_frames[b]) : true;
    float* yp = a.y + (int64_t)n * a.Cout * Ho * Wo + pix;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int co = 0; co < kMaxC; ++co) {
        if (co < a.Cout) {
            const float r = in ? fmaxf(acc[co], 0.f) : 0.f;
            if (in) yp[(int64_t)co * Ho * Wo] = r;
            if (a.stats) {
                const float s1 = wave_sum(r), s2 = wave_sum(r * r);
                if (lane == 0) { red[wave][2 * co] = s1; red[wave][2 * co + 1] = s2; }
            }
        }
    }
    if (a.stats && valid) {
        __syncthreads();
        if (threadIdx.x < 2 * a.Cout) {
            const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            atomicAdd(a.stats + ((int64_t)b * a.Cout * 2 + threadIdx.x), (double)s);
        }
    }
}

struct BnFinK {
    const double* stats;       // [B][C][2]
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;    // nullable (eval uses them as the source)
    float* mean; float* invstd;                 // [B][C] saved for backward
    float* scale; float* shift;                 // [B][C]  z = y*scale + shift
    const int32_t* n_frames;
    int B, T, C, HW; float eps, momentum; int training;
};

// one block per channel; thread b handles question b, thread 0 then advances the running statistics
// once per question in batch order (what B sequential reference calls would do)
__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFinK a) {
    const int c = blockIdx.x;
    for (int b = threadIdx.x; b < a.B; b += 256) {
        float mean, var;
        const double cnt = (double)(a.n_frames ? a.n_frames[b] : a.T) * a.HW;
        if (a.training) {
            const double s1 = a.stats[((int64_t)b * a.C + c) * 2], s2 = a.stats[((int64_t)b * a.C + c) * 2 + 1];
            const double m = cnt > 0 ? s1 / cnt : 0.0;
            mean = (float)m;
            var = cnt > 0 ? (float)fmax(s2 / cnt - m * m, 0.0) : 0.f;
        } else {
            mean = a.running_mean[c];
            var = a.running_var[c];
        }
        const float inv = cnt > 0 || !a.training ? 1.0f / sqrtf(var + a.eps) : 0.f;
        const int64_t i = (int64_t)b * a.C + c;
        a.mean[i] = mean; a.invstd[i] = inv;
        a.scale[i] = a.gamma[c] * inv;
        a.shift[i] = a.beta[c] - mean * a.gamma[c] * inv;
    }
    __syncthreads();
    if (threadIdx.x == 0 && a.training && a.running_mean) {
        float rm = a.running_mean[c], rv = a.running_var[c];
        for (int b = 0; b < a.B; ++b) {
            const double cnt = (double)(a.n_frames ? a.n_frames[b] : a.T) * a.HW;
            if (cnt <= 1) continue;
            const double s1 = a.stats[((int64_t)b * a.C + c) * 2], s2 = a.stats[((int64_t)b * a.C + c) * 2 + 1];
            const double m = s1 / cnt, v = fmax(s2 / cnt - m * m, 0.0) * cnt / (cnt - 1);   // unbiased for the running var
            rm = (1.f - a.momentum) * rm + a.momentum * (float)m;
            rv = (1.f - a.momentum) * rv + a.momentum * (float)v;
        }
        a.running_mean[c] = rm; a.running_var[c] = rv;
    }
}

struct BnApplyK {
    const float* y;            // [N][C][Hy][Wy]
    const float* scale; const float* shift;     // [B][C]
    float* z;                  // [N][C][Hz][Wz]
    uint8_t* argmax;           // [N][C][Hz][Wz] window position of the maximum (pool only, nullable)
    const int32_t* n_frames;
    int N, T, C, Hy, Wy, pool, tm;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyK a) {
    const int Hz = a.pool ? a.Hy / 3 : a.Hy, Wz = a.pool ? a.Wy / 3 : a.Wy;
    const int64_t total = (int64_t)a.N * a.C * Hz * Wz;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int px = (int)(i % Wz), py = (int)((i / Wz) % Hz), c = (int)((i / ((int64_t)Wz * Hz)) % a.C);
    const int n = (int)(i / ((int64_t)Wz * Hz * a.C));
    int b, t; frame_bt(n, a.N, a.T, a.tm, b, t);
    const bool valid = a.n_frames ? (t < a.n_frames[b]) : true;
    if (!valid) { a.z[i] = 0.f; if (a.argmax) a.argmax[i] = 0; return; }
    const float sc = a.scale[(int64_t)b * a.C + c], sh = a.shift[(int64_t)b * a.C + c];
    const float* yp = a.y + ((int64_t)n * a.C + c) * a.Hy * a.Wy;
    if (!a.pool) { a.z[i] = yp[(int64_t)py * a.Wy + px] * sc + sh; return; }
    float best = -INFINITY; int bi = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {          // row-major scan, first maximum wins (torch max_pool2d)
        const float v = yp[(int64_t)(py * 3 + k / 3) * a.Wy + px * 3 + k % 3] * sc + sh;
        if (v > best) { best = v; bi = k; }
    }
    a.z[i] = best;
    if (a.argmax) a.argmax[i] = (uint8_t)bi;
}

struct BnBwdK {
    const float* y;            // [N][C][Hy][Wy] relu(conv)
    const float* dz;           // [N][C][Hz][Wz]
    const uint8_t* argmax;     // pool only
    const float* mean; const float* invstd;     // [B][C]
    const float* gamma;
    double* sums;              // [B][C][2]  sum dz', sum dz'*xhat
    float* dgamma; float* dbeta;                // [C] accumulate
    float* dconv;              // [N][C][Hy][Wy] out
    const int32_t* n_frames;
    int N, T, C, Hy, Wy, pool, B, tm;
};

// per (question, channel) sums over the pooled domain: every pooled gradient reaches exactly one y element
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdK a) {
    __shared__ float red[4][2];
    const int Hz = a.pool ? a.Hy / 3 : a.Hy, Wz = a.pool ? a.Wy / 3 : a.Wy;
    const int n = blockIdx.z, c = blockIdx.y;
    int b, t; frame_bt(n, a.N, a.T, a.tm, b, t);
    const bool valid = a.n_frames ? (t < a.n_frames[b]) : true;
    if (!valid) return;
    const float mu = a.mean[(int64_t)b * a.C + c], inv = a.invstd[(int64_t)b * a.C + c];
    const float* yp = a.y + ((int64_t)n * a.C + c) * a.Hy * a.Wy;
    const float* dp = a.dz + ((int64_t)n * a.C + c) * Hz * Wz;
    const uint8_t* ap = a.argmax ? a.argmax + ((int64_t)n * a.C + c) * Hz * Wz : nullptr;
    float s1 = 0.f, s2 = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hz * Wz; i += gridDim.x * 256) {
        const float d = dp[i];
        int yy = i / Wz, xx = i % Wz;
        if (a.pool) { const int k = ap[i]; yy = yy * 3 + k / 3; xx = xx * 3 + k % 3; }
        const float xh = (yp[(int64_t)yy * a.Wy + xx] - mu) * inv;
        s1 += d; s2 += d * xh;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(a.sums + ((int64_t)b * a.C + c) * 2 + threadIdx.x, (double)s);
        atomicAdd((threadIdx.x == 0 ? a.dbeta : a.dgamma) + c, s);
    }
}

// dconv = relu'(y) * gamma*invstd * (dz' - mean(dz') - xhat * mean(dz' xhat)), dz' = routed pooled gradient
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(BnBwdK a) {
    const int Hz = a.pool ? a.Hy / 3 : a.Hy, Wz = a.pool ? a.Wy / 3 : a.Wy;
    const int64_t total = (int64_t)a.N * a.C * a.Hy * a.Wy;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % a.Wy), yq = (int)((i / a.Wy) % a.Hy), c = (int)((i / ((int64_t)a.Wy * a.Hy)) % a.C);
    const int n = (int)(i / ((int64_t)a.Wy * a.Hy * a.C));
    int b, t; frame_bt(n, a.N, a.T, a.tm, b, t);
    const int nf = a.n_frames ? a.n_frames[b] : a.T;
    if (t >= nf) { a.dconv[i] = 0.f; return; }
    const float yv = a.y[i];
    float d = 0.f;
    if (a.pool) {
        const int py = yq / 3, px = x / 3;
        if (py < Hz && px < Wz) {
            const int64_t pi = ((int64_t)n * a.C + c) * Hz * Wz + (int64_t)py * Wz + px;
            if (a.argmax[pi] == (yq % 3) * 3 + (x % 3)) d = a.dz[pi];
        }
    } else {
        d = a.dz[i];
    }
    const int64_t bc = (int64_t)b * a.C + c;
    const float inv = a.invstd[bc];
    const double cnt = (double)nf * a.Hy * a.Wy;
    const float m1 = (float)(a.sums[bc * 2] / cnt), m2 = (float)(a.sums[bc * 2 + 1] / cnt);
    const float xh = (yv - a.mean[bc]) * inv;
    const float dy = a.gamma[c] * inv * (d - m1 - xh * m2);
    a.dconv[i] = yv > 0.f ? dy : 0.f;
}

struct ConvBwdK {
    const float* x;            // [N][Cin][H][W] block input
    const float* w;            // [Cout][Cin][3][3]
    const float* dconv;        // [N][Cout][H-2][W-2]
    float* dx;                 // [N][Cin][H][W] (input-gradient kernel)
    float* dw; float* dbias;   // accumulate (weight-gradient kernel)
    int N, Cin, Cout, H, W;
};

__global__ __launch_bounds__(256) void conv3x3_bwd_input_kernel(ConvBwdK a) {
    __shared__ float ws[kMaxC * kMaxC * 9];
    const int n = blockIdx.y, Ho = a.H - 2, Wo = a.W - 2;
    for (int i = threadIdx.x; i < a.Cout * a.Cin * 9; i += 256) ws[i] = a.w[i];
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= a.H * a.W) return;
    const int yy = pix / a.W, xx = pix % a.W;
    float acc[kMaxC];
#pragma unroll
    for (int ci = 0; ci < kMaxC; ++ci) acc[ci] = 0.f;
    for (int co = 0; co < a.Cout; ++co) {
        const float* dp = a.dconv + ((int64_t)n * a.Cout + co) * Ho * Wo;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int oy = yy - dy;
            if (oy < 0 || oy >= Ho) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ox = xx - dx;
                if (ox < 0 || ox >= Wo) continue;
                const float d = dp[(int64_t)oy * Wo + ox];
#pragma unroll
                for (int ci = 0; ci < kMaxC; ++ci)
                    if (ci < a.Cin) acc[ci] += d * ws[(co * a.Cin + ci) * 9 + dy * 3 + dx];
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < kMaxC; ++ci)
        if (ci < a.Cin) a.dx[((int64_t)n * a.Cin + ci) * a.H * a.W + pix] = acc[ci];
}

// workgroups over (frame, output channel, pixel slice): every thread keeps the Cin*9 tap products of its
// pixels, block reduction, then one atomic per tap
__global__ __launch_bounds__(256) void conv3x3_bwd_weight_kernel(ConvBwdK a) {
    __shared__ float red[4][kMaxC * 9 + 1];
    const int n = blockIdx.x, co = blockIdx.y, Ho = a.H - 2, Wo = a.W - 2;
    const float* dp = a.dconv + ((int64_t)n * a.Cout + co) * Ho * Wo;
    const float* xp = a.x + (int64_t)n * a.Cin * a.H * a.W;
    float acc[kMaxC * 9];
#pragma unroll
    for (int k = 0; k < kMaxC * 9; ++k) acc[k] = 0.f;
    float db = 0.f;
    for (int pix = blockIdx.z * 256 + threadIdx.x; pix < Ho * Wo; pix += gridDim.z * 256) {
        const float d = dp[pix];
        db += d;
        const int oy = pix / Wo, ox = pix % Wo;
#pragma unroll
        for (int ci = 0; ci < kMaxC; ++ci) {
            if (ci < a.Cin) {
                const float* q = xp + (int64_t)ci * a.H * a.W + (int64_t)oy * a.W + ox;
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[ci * 9 + k] += d * q[(k / 3) * a.W + (k % 3)];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kMaxC * 9; ++k) {
        if (k < a.Cin * 9) {
            const float s = wave_sum(acc[k]);
            if (lane == 0) red[wave][k] = s;
        }
    }
    db = wave_sum(db);
    if (lane == 0) red[wave][kMaxC * 9] = db;
    __syncthreads();
    for (int k = threadIdx.x; k < a.Cin * 9; k += 256)
        atomicAdd(a.dw + (int64_t)co * a.Cin * 9 + k, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
    if (threadIdx.x == 0)
        atomicAdd(a.dbias + co, red[0][kMaxC * 9] + red[1][kMaxC * 9] + red[2][kMaxC * 9] + red[3][kMaxC * 9]);
}

__global__ __launch_bounds__(256) void zero_f64_kernel(double* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

int check_cnn(const mmqg_frame_cnn& d, const char* who) {
    MMQG_REQUIRE(d.B >= 0 && d.T > 0 && d.H >= 3 && d.W >= 3, "%s: bad shape", who);
    MMQG_REQUIRE(d.n_blocks >= 1 && d.n_blocks <= MMQG_CNN_MAX_BLOCKS, "%s: 1..%d blocks", who, MMQG_CNN_MAX_BLOCKS);
    MMQG_REQUIRE(d.frames, "%s: null frames", who);
    int h = d.H, w = d.W, cin = d.Cin;
    for (int i = 0; i < d.n_blocks; ++i) {
        const mmqg_cnn_block& b = d.block[i];
        MMQG_REQUIRE(b.cout >= 1 && b.cout <= kMaxC && cin >= 1 && cin <= kMaxC, "%s: channel counts must be in [1,%d]", who, kMaxC);
        MMQG_REQUIRE(b.w && b.bias && b.gamma && b.beta && b.y && b.z && b.stats && b.mean && b.invstd && b.scale && b.shift,
                     "%s: null buffer in block %d", who, i);
        MMQG_REQUIRE(!b.pool || b.argmax, "%s: pooled block %d needs the argmax buffer", who, i);
        h -= 2; w -= 2;
        MMQG_REQUIRE(h >= 1 && w >= 1, "%s: image too small for block %d", who, i);
        if (b.pool) { h /= 3; w /= 3; MMQG_REQUIRE(h >= 1 && w >= 1, "%s: image too small for the pool of block %d", who, i); }
        cin = b.cout;
    }
    return 0;
}

}  // namespace

namespace mmqg {

int frame_cnn_fwd(const mmqg_frame_cnn& d, hipStream_t s) {
    MMQG_TRY(check_cnn(d, "frame_cnn_fwd"));
    if (d.B == 0) return 0;
    const int N = d.B * d.T;
    const float* x = d.frames;
    int h = d.H, w = d.W, cin = d.Cin;
    for (int i = 0; i < d.n_blocks; ++i) {
        const mmqg_cnn_block& b = d.block[i];
        const int ho = h - 2, wo = w - 2;
        if (d.training) {
            const int64_t ns = (int64_t)d.B * b.cout * 2;
            hipLaunchKernelGGL(zero_f64_kernel, dim3((unsigned)ceil_div64(ns, 256)), dim3(256), 0, s, b.stats, ns);
        }
        ConvK c{x, b.w, b.bias, b.y, d.training ? b.stats : nullptr, d.n_frames, N, d.T, cin, b.cout, h, w, d.time_major};
        hipLaunchKernelGGL(conv3x3_relu_stats_kernel, dim3(ceil_div(ho * wo, 256), N), dim3(256), 0, s, c);
        MMQG_TRY(check_launch("conv3x3_relu_stats"));
        BnFinK f{b.stats, b.gamma, b.beta, b.running_mean, b.running_var, b.mean, b.invstd, b.scale, b.shift, d.n_frames,
                 d.B, d.T, b.cout, ho * wo, d.eps, d.momentum, d.training};
        MMQG_REQUIRE(d.training || (b.running_mean && b.running_var), "frame_cnn_fwd: eval mode needs running statistics");
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(b.cout), dim3(256), 0, s, f);
        MMQG_TRY(check_launch("bn_finalize"));
        const int hz = b.pool ? ho / 3 : ho, wz = b.pool ? wo / 3 : wo;
        BnApplyK ap{b.y, b.scale, b.shift, b.z, b.pool ? b.argmax : nullptr, d.n_frames, N, d.T, b.cout, ho, wo, b.pool, d.time_major};
        hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div64((int64_t)N * b.cout * hz * wz, 256)), dim3(256), 0, s, ap);
        MMQG_TRY(check_launch("bn_apply"));
        x = b.z; h = hz; w = wz; cin = b.cout;
    }
    return 0;
}

int frame_cnn_bwd(const mmqg_frame_cnn& d, const mmqg_frame_cnn_grad& g, hipStream_t s) {
    MMQG_TRY(check_cnn(d, "frame_cnn_bwd"));
    if (d.B == 0) return 0;
    MMQG_REQUIRE(d.training, "frame_cnn_bwd: backward is defined for training mode (batch statistics)");
    MMQG_REQUIRE(g.dfeat && g.dconv && g.dz, "frame_cnn_bwd: null buffer");
    const int N = d.B * d.T;
    int hs[MMQG_CNN_MAX_BLOCKS + 1], wsz[MMQG_CNN_MAX_BLOCKS + 1], cs[MMQG_CNN_MAX_BLOCKS + 1];
    hs[0] = d.H; wsz[0] = d.W; cs[0] = d.Cin;
    for (int i = 0; i < d.n_blocks; ++i) {
        const int ho = hs[i] - 2, wo = wsz[i] - 2;
        hs[i + 1] = d.block[i].pool ? ho / 3 : ho;
        wsz[i + 1] = d.block[i].pool ? wo / 3 : wo;
        cs[i + 1] = d.block[i].cout;
    }
    const float* dz = g.dfeat;          // gradient of the last block's output == the flattened features
    for (int i = d.n_blocks - 1; i >= 0; --i) {
        const mmqg_cnn_block& b = d.block[i];
        MMQG_REQUIRE(g.dw[i] && g.dbias[i] && g.dgamma[i] && g.dbeta[i], "frame_cnn_bwd: null gradient (block %d)", i);
        const int h = hs[i], w = wsz[i], cin = cs[i], ho = h - 2, wo = w - 2;
        const int hz = hs[i + 1], wz = wsz[i + 1];
        const int64_t ns = (int64_t)d.B * b.cout * 2;
        hipLaunchKernelGGL(zero_f64_kernel, dim3((unsigned)ceil_div64(ns, 256)), dim3(256), 0, s, b.stats, ns);
        BnBwdK k{b.y, dz, b.pool ? b.argmax : nullptr, b.mean, b.invstd, b.gamma, b.stats, g.dgamma[i], g.dbeta[i], g.dconv,
                 d.n_frames, N, d.T, b.cout, ho, wo, b.pool, d.B, d.time_major};
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(std::min(8, ceil_div(hz * wz, 256)), b.cout, N), dim3(256), 0, s, k);
        MMQG_TRY(check_launch("bn_bwd_reduce"));
        hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3((unsigned)ceil_div64((int64_t)N * b.cout * ho * wo, 256)), dim3(256), 0, s, k);
        MMQG_TRY(check_launch("bn_relu_bwd"));
        const float* xin = i == 0 ? d.frames : d.block[i - 1].z;
        ConvBwdK cb{xin, b.w, g.dconv, g.dz, g.dw[i], g.dbias[i], N, cin, b.cout, h, w};
        // enough pixel slices to put ~2k workgroups on the chip, each with at least 4 pixels per thread
        const int slices = std::max(1, std::min(ceil_div(ho * wo, 1024), ceil_div(2048, N * b.cout)));
        hipLaunchKernelGGL(conv3x3_bwd_weight_kernel, dim3(N, b.cout, slices), dim3(256), 0, s, cb);
        MMQG_TRY(check_launch("conv3x3_bwd_weight"));
        if (i > 0) {
            hipLaunchKernelGGL(conv3x3_bwd_input_kernel, dim3(ceil_div(h * w, 256), N), dim3(256), 0, s, cb);
            MMQG_TRY(check_launch("conv3x3_bwd_input"));
            dz = g.dz;
        }
    }
    return 0;
}

}  // namespace mmqg
