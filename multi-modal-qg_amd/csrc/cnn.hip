// Frame CNN of VideoConvLstmEncoder (model/encoder.py:40-50,64-67): four blocks of
//   3x3 valid convolution (stride 1) -> ReLU -> BatchNorm2d, with a 3x3/3 max-pool after blocks 2 and 4.
// Channel counts are tiny (3 -> 4 -> 6 -> 8 -> 10), so this is memory-bound element-wise work, not a
// GEMM: direct convolution.  For the reference's channel counts a thread produces a 1x4 vertical strip
// of outputs for every output channel (6x3 input patch per input channel loaded once, every load
// coalesced along x; weights at compile-time offsets through the scalar cache); other channel counts
// take generic one-pixel-per-thread kernels with the weights in LDS.  BatchNorm statistics are per QUESTION (the reference feeds one question's T frames as the
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

// Specialised convolution for compile-time channel counts.  !BWD: y = relu(conv(x) + bias) and the
// per-question statistics; reads CA = Cin channels of [H][W], writes CB = Cout channels of [H-2][W-2].
// BWD: dx = full correlation of dconv with the flipped kernels; reads CA = Cout channels of
// [H-2][W-2] (through a.x), writes CB = Cin channels of [H][W] (through a.y).
template <int CA, int CB, bool BWD>
__global__ __launch_bounds__(256) void conv3x3_strip_kernel(ConvK a) {
    __shared__ float red[4][2 * CB];
    const int Hi = BWD ? a.H - 2 : a.H, Wi = BWD ? a.W - 2 : a.W;
    const int Hout = BWD ? a.H : a.H - 2, Wout = BWD ? a.W : a.W - 2;
    const int off = BWD ? -2 : 0;
    const int n = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool inb = g < ((Hout + 3) >> 2) * Wout;
    const int ox = inb ? g % Wout : 0, oy0 = inb ? (g / Wout) * 4 : 0;
    const float* __restrict__ w = a.w;
    float acc[4][CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const float b0 = BWD ? 0.f : a.bias[cb];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][cb] = b0;
    }
    const float* __restrict__ xin = a.x + (int64_t)n * CA * Hi * Wi;
#pragma unroll
    for (int ca = 0; ca < CA; ++ca) {
        float v[6][3];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = oy0 + r + off;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ix = ox + c + off;
                const bool ok = inb && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
                v[r][c] = ok ? xin[((int64_t)ca * Hi + iy) * Wi + ix] : 0.f;
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float wv = BWD ? w[((ca * CB + cb) * 3 + (2 - dy)) * 3 + (2 - dx)] : w[((cb * CA + ca) * 3 + dy) * 3 + dx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][cb] += v[j + dy][dx] * wv;
                }
            }
        }
    }
    float* __restrict__ yout = a.y + (int64_t)n * CB * Hout * Wout;
    if (BWD) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (inb && oy0 + j < Hout) yout[((int64_t)cb * Hout + oy0 + j) * Wout + ox] = acc[j][cb];
        return;
    }
    int b, t; frame_bt(n, a.N, a.T, a.tm, b, t);
    const bool valid = a.n_frames ? (t < a.n_frames[b]) : true;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (inb && oy0 + j < Hout) {
                const float r = fmaxf(acc[j][cb], 0.f);
                yout[((int64_t)cb * Hout + oy0 + j) * Wout + ox] = r;
                s1 += r; s2 += r * r;
            }
        }
        if (a.stats) {
            s1 = wave_sum(s1); s2 = wave_sum(s2);
            if (lane == 0) { red[wave][2 * cb] = s1; red[wave][2 * cb + 1] = s2; }
        }
    }
    if (a.stats && valid) {
        __syncthreads();
        if (threadIdx.x < 2 * CB) {
            const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            atomicAdd(a.stats + ((int64_t)b * CB * 2 + threadIdx.x), (double)s);
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

// one block per channel; thread b handles question b (statistics parked in LDS), thread 0 then advances
// the running statistics once per question in batch order (what B sequential reference calls would do)
__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFinK a) {
    __shared__ float q_mean[256], q_var[256];    // batch mean / unbiased variance of question b (chunk of 256)
    __shared__ int q_ok[256];
    const int c = blockIdx.x;
    float rm = 0.f, rv = 0.f;
    const bool ema = a.training && a.running_mean;
    if (threadIdx.x == 0 && ema) { rm = a.running_mean[c]; rv = a.running_var[c]; }
    for (int b0 = 0; b0 < a.B; b0 += 256) {
        const int b = b0 + threadIdx.x;
        if (b < a.B) {
            float mean, var;
            const double cnt = (double)(a.n_frames ? a.n_frames[b] : a.T) * a.HW;
            q_ok[threadIdx.x] = 0;
            if (a.training) {
                const double s1 = a.stats[((int64_t)b * a.C + c) * 2], s2 = a.stats[((int64_t)b * a.C + c) * 2 + 1];
                const double m = cnt > 0 ? s1 / cnt : 0.0;
                const double v = cnt > 0 ? fmax(s2 / cnt - m * m, 0.0) : 0.0;
                mean = (float)m;
                var = (float)v;
                if (cnt > 1) { q_ok[threadIdx.x] = 1; q_mean[threadIdx.x] = mean; q_var[threadIdx.x] = (float)(v * cnt / (cnt - 1)); }
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
        if (threadIdx.x == 0 && ema) {
            const int nb = min(256, a.B - b0);
            for (int j = 0; j < nb; ++j) {
                if (!q_ok[j]) continue;
                rm = (1.f - a.momentum) * rm + a.momentum * q_mean[j];
                rv = (1.f - a.momentum) * rv + a.momentum * q_var[j];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && ema) { a.running_mean[c] = rm; a.running_var[c] = rv; }
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
    const int pix = blockIdx.y * 256 + threadIdx.x;             // grid: (frame*channel planes, pixel blocks)
    if (pix >= Hz * Wz) return;
    const int n = blockIdx.x / a.C, c = blockIdx.x % a.C;
    const int px = pix % Wz, py = pix / Wz;
    const int64_t i = (int64_t)blockIdx.x * Hz * Wz + pix;
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

// per (question, channel) sums over the pooled domain: every pooled gradient reaches exactly one y element.
// grid (pixel slices, C, B): a workgroup walks the question's valid frames, so only `slices` workgroups
// meet on each accumulator
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdK a) {
    __shared__ float red[4][2];
    const int Hz = a.pool ? a.Hy / 3 : a.Hy, Wz = a.pool ? a.Wy / 3 : a.Wy;
    const int b = blockIdx.z, c = blockIdx.y;
    const int nf = a.n_frames ? min(a.n_frames[b], a.T) : a.T;
    const float mu = a.mean[(int64_t)b * a.C + c], inv = a.invstd[(int64_t)b * a.C + c];
    float s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < nf; ++t) {
        const int n = a.tm ? t * a.B + b : b * a.T + t;
        const float* yp = a.y + ((int64_t)n * a.C + c) * a.Hy * a.Wy;
        const float* dp = a.dz + ((int64_t)n * a.C + c) * Hz * Wz;
        const uint8_t* ap = a.argmax ? a.argmax + ((int64_t)n * a.C + c) * Hz * Wz : nullptr;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < Hz * Wz; i += gridDim.x * 256) {
            const float d = dp[i];
            int yy = i / Wz, xx = i % Wz;
            if (a.pool) { const int k = ap[i]; yy = yy * 3 + k / 3; xx = xx * 3 + k % 3; }
            const float xh = (yp[(int64_t)yy * a.Wy + xx] - mu) * inv;
            s1 += d; s2 += d * xh;
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(a.sums + ((int64_t)b * a.C + c) * 2 + threadIdx.x, (double)s);
        atomicAdd((threadIdx.x == 0 ? a.dbeta : a.dgamma) + c, s);      // slices*B adds per channel
    }
}

// dconv = relu'(y) * gamma*invstd * (dz' - mean(dz') - xhat * mean(dz' xhat)), dz' = routed pooled gradient
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(BnBwdK a) {
    __shared__ float coef[5];      // per (question, channel): mean, invstd, gamma*invstd, mean(dz'), mean(dz' xhat)
    __shared__ int frame_valid;
    const int Hz = a.pool ? a.Hy / 3 : a.Hy, Wz = a.pool ? a.Wy / 3 : a.Wy;
    const int n = blockIdx.x / a.C, c = blockIdx.x % a.C;        // grid: (frame*channel planes, pixel blocks)
    if (threadIdx.x == 0) {        // the plane's coefficients once per workgroup (two f64 divisions), not per pixel
        int b, t; frame_bt(n, a.N, a.T, a.tm, b, t);
        const int nf = a.n_frames ? a.n_frames[b] : a.T;
        frame_valid = t < nf;
        const int64_t bc = (int64_t)b * a.C + c;
        const double cnt = (double)nf * a.Hy * a.Wy;
        const float inv = a.invstd[bc];
        coef[0] = a.mean[bc]; coef[1] = inv; coef[2] = a.gamma[c] * inv;
        coef[3] = cnt > 0 ? (float)(a.sums[bc * 2] / cnt) : 0.f;
        coef[4] = cnt > 0 ? (float)(a.sums[bc * 2 + 1] / cnt) : 0.f;
    }
    __syncthreads();
    const int pix = blockIdx.y * 256 + threadIdx.x;
    if (pix >= a.Hy * a.Wy) return;
    const int x = pix % a.Wy, yq = pix / a.Wy;
    const int64_t i = (int64_t)blockIdx.x * a.Hy * a.Wy + pix;
    if (!frame_valid) { a.dconv[i] = 0.f; return; }
    const float yv = a.y[i];
    float d = 0.f;
    if (a.pool) {
        const int py = yq / 3, px = x / 3;
        if (py < Hz && px < Wz) {
            const int64_t pi = (int64_t)blockIdx.x * Hz * Wz + (int64_t)py * Wz + px;
            if (a.argmax[pi] == (yq - 3 * py) * 3 + (x - 3 * px)) d = a.dz[pi];
        }
    } else {
        d = a.dz[i];
    }
    const float xh = (yv - coef[0]) * coef[1];
    const float dy = coef[2] * (d - coef[3] - xh * coef[4]);
    a.dconv[i] = yv > 0.f ? dy : 0.f;
}

struct ConvBwdK {
    const float* x;            // [N][Cin][H][W] block input
    const float* w;            // [Cout][Cin][3][3]
    const float* dconv;        // [N][Cout][H-2][W-2]
    float* dx;                 // [N][Cin][H][W] (input-gradient kernel)
    float* dw; float* dbias;   // accumulate (weight-gradient kernel)
    int N, Cin, Cout, H, W;
    const int32_t* n_frames; int T, tm;
    int fpw;                   // frames per workgroup (specialised weight-gradient kernel)
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

// Specialised weight gradient: workgroup = (a.fpw consecutive frames, group of G output channels); a thread
// walks 1x4 vertical strips of those frames, loading the 6x3 input patch of each input channel once for the strip's
// 4 pixels x G channels x 9 taps; block reduction, one atomic per tap.  Padding frames are skipped.
template <int CI, int CO, int G>
__global__ __launch_bounds__(256) void conv3x3_wgrad_strip_kernel(ConvBwdK a) {
    constexpr int NA = G * CI * 9;
    __shared__ float red[4][NA + G];
    const int co0 = blockIdx.y * G, Ho = a.H - 2, Wo = a.W - 2;
    float acc[G][CI][9];
    float db[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        db[q] = 0.f;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[q][ci][k] = 0.f;
    }
    const int groups = ((Ho + 3) >> 2) * Wo;
    for (int idx = threadIdx.x; idx < a.fpw * groups; idx += 256) {
        const int n = blockIdx.x * a.fpw + idx / groups, g = idx % groups;
        if (n >= a.N) break;
        int b, t; frame_bt(n, a.N, a.T, a.tm, b, t);
        if (a.n_frames && t >= a.n_frames[b]) continue;
        const float* __restrict__ dp = a.dconv + ((int64_t)n * CO + co0) * Ho * Wo;
        const float* __restrict__ xp = a.x + (int64_t)n * CI * a.H * a.W;
        const int ox = g % Wo, oy0 = (g / Wo) * 4;
        float d[4][G];
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[j][q] = (co0 + q < CO && oy0 + j < Ho) ? dp[((int64_t)q * Ho + oy0 + j) * Wo + ox] : 0.f;
                db[q] += d[j][q];
            }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            float v[6][3];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v[r][c] = (oy0 + r < a.H) ? xp[((int64_t)ci * a.H + oy0 + r) * a.W + ox + c] : 0.f;
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[q][ci][dy * 3 + dx] += d[j][q] * v[j + dy][dx];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < G; ++q) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float sum = wave_sum(acc[q][ci][k]);
                if (lane == 0) red[wave][(q * CI + ci) * 9 + k] = sum;
            }
        const float sb = wave_sum(db[q]);
        if (lane == 0) red[wave][NA + q] = sb;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NA + G; k += 256) {
        const float sum = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        const int q = k < NA ? k / (CI * 9) : k - NA;
        if (co0 + q >= CO) continue;
        if (k < NA) atomicAdd(a.dw + (int64_t)co0 * CI * 9 + k, sum);
        else atomicAdd(a.dbias + co0 + q, sum);
    }
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

// the reference's channel pairs (encoder.py:40-49) get the specialised kernels
bool launch_conv_fwd(const ConvK& c, int pix_blocks_strip, hipStream_t s) {
#define MMQG_CASE(CA, CB)                                                                                              \
    if (c.Cin == CA && c.Cout == CB) {                                                                                 \
        hipLaunchKernelGGL((conv3x3_strip_kernel<CA, CB, false>), dim3(pix_blocks_strip, c.N), dim3(256), 0, s, c);    \
        return true;                                                                                                   \
    }
    MMQG_CASE(3, 4) MMQG_CASE(4, 6) MMQG_CASE(6, 8) MMQG_CASE(8, 10)
#undef MMQG_CASE
    return false;
}

bool launch_conv_bwd_input(const ConvBwdK& k, hipStream_t s) {
    // the strip kernel reads a.x = dconv (Cout channels of [H-2][W-2]) and writes a.y = dx (Cin channels of [H][W])
    ConvK c{k.dconv, k.w, nullptr, k.dx, nullptr, nullptr, k.N, 1, k.Cout, k.Cin, k.H, k.W, 0};
    const int blocks = mmqg::ceil_div(((k.H + 3) / 4) * k.W, 256);
#define MMQG_CASE(CA, CB)                                                                                              \
    if (k.Cout == CA && k.Cin == CB) {                                                                                 \
        hipLaunchKernelGGL((conv3x3_strip_kernel<CA, CB, true>), dim3(blocks, k.N), dim3(256), 0, s, c);               \
        return true;                                                                                                   \
    }
    MMQG_CASE(6, 4) MMQG_CASE(8, 6) MMQG_CASE(10, 8)
#undef MMQG_CASE
    return false;
}

bool launch_conv_wgrad(ConvBwdK k, hipStream_t s) {
    // small images: several frames per workgroup (about 8 strips per thread) so the block reduction of
    // the G*Cin*9 accumulators is amortised
    const int groups = ((k.H - 2 + 3) / 4) * (k.W - 2);
    k.fpw = std::max(1, std::min(8, 2048 / std::max(groups, 1)));
    const int wgx = mmqg::ceil_div(k.N, k.fpw);
#define MMQG_CASE(CI, CO, G)                                                                                           \
    if (k.Cin == CI && k.Cout == CO) {                                                                                 \
        hipLaunchKernelGGL((conv3x3_wgrad_strip_kernel<CI, CO, G>), dim3(wgx, (CO + G - 1) / G), dim3(256), 0, s, k);  \
        return true;                                                                                                   \
    }
    MMQG_CASE(3, 4, 4) MMQG_CASE(4, 6, 3) MMQG_CASE(6, 8, 2) MMQG_CASE(8, 10, 2)
#undef MMQG_CASE
    return false;
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
        if (!launch_conv_fwd(c, ceil_div(((ho + 3) / 4) * wo, 256), s))
            hipLaunchKernelGGL(conv3x3_relu_stats_kernel, dim3(ceil_div(ho * wo, 256), N), dim3(256), 0, s, c);
        MMQG_TRY(check_launch("conv3x3_relu_stats"));
        BnFinK f{b.stats, b.gamma, b.beta, b.running_mean, b.running_var, b.mean, b.invstd, b.scale, b.shift, d.n_frames,
                 d.B, d.T, b.cout, ho * wo, d.eps, d.momentum, d.training};
        MMQG_REQUIRE(d.training || (b.running_mean && b.running_var), "frame_cnn_fwd: eval mode needs running statistics");
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(b.cout), dim3(256), 0, s, f);
        MMQG_TRY(check_launch("bn_finalize"));
        const int hz = b.pool ? ho / 3 : ho, wz = b.pool ? wo / 3 : wo;
        BnApplyK ap{b.y, b.scale, b.shift, b.z, b.pool ? b.argmax : nullptr, d.n_frames, N, d.T, b.cout, ho, wo, b.pool, d.time_major};
        hipLaunchKernelGGL(bn_apply_kernel, dim3(N * b.cout, ceil_div(hz * wz, 256)), dim3(256), 0, s, ap);
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
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(std::min(8, ceil_div(hz * wz, 1024)), b.cout, d.B), dim3(256), 0, s, k);
        MMQG_TRY(check_launch("bn_bwd_reduce"));
        hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(N * b.cout, ceil_div(ho * wo, 256)), dim3(256), 0, s, k);
        MMQG_TRY(check_launch("bn_relu_bwd"));
        const float* xin = i == 0 ? d.frames : d.block[i - 1].z;
        ConvBwdK cb{xin, b.w, g.dconv, g.dz, g.dw[i], g.dbias[i], N, cin, b.cout, h, w, d.n_frames, d.T, d.time_major, 1};
        if (!launch_conv_wgrad(cb, s)) {
            // enough pixel slices to put ~2k workgroups on the chip, each with at least 4 pixels per thread
            const int slices = std::max(1, std::min(ceil_div(ho * wo, 1024), ceil_div(2048, N * b.cout)));
            hipLaunchKernelGGL(conv3x3_bwd_weight_kernel, dim3(N, b.cout, slices), dim3(256), 0, s, cb);
        }
        MMQG_TRY(check_launch("conv3x3_bwd_weight"));
        if (i > 0) {
            if (!launch_conv_bwd_input(cb, s))
                hipLaunchKernelGGL(conv3x3_bwd_input_kernel, dim3(ceil_div(h * w, 256), N), dim3(256), 0, s, cb);
            MMQG_TRY(check_launch("conv3x3_bwd_input"));
            dz = g.dz;
        }
    }
    return 0;
}

}  // namespace mmqg
