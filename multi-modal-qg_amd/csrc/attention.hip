// Decoder attention for one time step (model/decoder.py:78-95): three location-style
// attentions (text / audio / video) whose scores have already been produced by the score
// GEMM.  This file holds the HBM-bound part:
//
//   fwd  : per (question, modality): softmax over the score segment, then
//          ctx[:] = sum_i w[i] * V[i][:]  — a weighted row sum that streams the question's
//          value rows exactly once.  0.5 FLOP per byte -> bounded by HBM bandwidth.
//   bwd  : d(attn)[i] = V[i][:] . dctx[:]  (streams the rows once more), then the softmax
//          Jacobian; the value gradient is NOT accumulated step by step: attn_dvalues forms
//          it once after the time loop as sum_t attn[t] (x) dctx[t], and only for the rows an
//          encoder actually produced.
//
// Work split.  The value tensor of a batch is B x (Lt*H + Lav*Da + Lav*Dv) floats
// (838,144 B per question at the config.py extents).  Forward: one workgroup per
// (question, modality, 128-column chunk); the 256 threads are 32 column lanes (float4 each,
// 512 contiguous bytes per row) x 8 row groups, every thread keeps 4 independent 16-byte
// loads in flight; the softmax weights of the segment live in LDS (the reductions are
// wavefront shuffles + one 4-entry LDS exchange), the 8 row-group partials are combined
// through LDS.  Backward: one workgroup per (question, modality, 32-row block), one
// wavefront per row at a time, lanes across the row's columns (whole contiguous rows),
// shuffle reduction per row.
#include <stdlib.h>

#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

constexpr int kMaxRows = 4096;   // longest score segment the LDS weight buffer holds
constexpr int kRowBlock = 32;    // value rows per backward workgroup

struct Segment {
    const float* base;   // value rows of question b for this modality
    int L, D;            // rows, columns
    int seg_off;         // offset of the segment inside a score row
    int ctx_off;         // offset inside a context row
    int valid;           // rows that count under MMQG_MASK_INTENDED
};

__device__ __forceinline__ Segment pick_segment(const mmqg_attn_values& v, int modality, int b) {
    Segment s;
    if (modality == 0) {
        s.base = v.text + (int64_t)b * v.text_stride_b; s.L = v.Lt; s.D = v.H; s.seg_off = 0; s.ctx_off = 0;
        s.valid = v.text_len ? v.text_len[b] : v.Lt;
    } else if (modality == 1) {
        s.base = v.audio + (int64_t)b * v.audio_stride_b; s.L = v.Lav; s.D = v.Da; s.seg_off = v.Lt; s.ctx_off = v.H;
        s.valid = v.av_len ? v.av_len[b] : v.Lav;
    } else {
        s.base = v.video + (int64_t)b * v.video_stride_b; s.L = v.Lav; s.D = v.Dv; s.seg_off = v.Lt + v.Lav;
        s.ctx_off = v.H + v.Da;
        s.valid = v.av_len ? v.av_len[b] : v.Lav;
    }
    return s;
}

__device__ __forceinline__ float block_max(float x, float* sh) {
    x = wave_max(x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = x;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sh[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float x, float* sh) {
    x = wave_sum(x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = x;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
    return r;
}

struct AttnFwdK {
    mmqg_attn_values v;
    const float* scores; int ld_s;
    float* attn; int ld_a;
    float* ctx; int ld_c;
    int chunks_text, chunks_audio, chunks_video;   // column chunks per question and modality
    int audio_chunk;                               // columns per audio chunk (32 when the main chunk is 64)
    int vec_text, vec_audio, vec_video;
};

// One work item: (question b, modality, chunk of kChunk value columns): 256 threads = (kChunk/4) float4 column lanes x
// row groups.
template <int kChunk>
__device__ __forceinline__ void attn_fwd_item(const AttnFwdK& a, int modality, int b, int chunk, float* w, float* red, float* sh) {
    constexpr int kLanes = kChunk / 4, kGroups = 256 / kLanes;
    const Segment sg = pick_segment(a.v, modality, b);
    const bool vec = modality == 0 ? a.vec_text : (modality == 1 ? a.vec_audio : a.vec_video);
    const int tid = threadIdx.x;
    const bool masked = a.v.mask_mode == MMQG_MASK_INTENDED;

    // ---- this thread's slice of the value stream: column lane cl (float4), row group rg
    const int cl = tid % kLanes, rg = tid / kLanes;
    const int col = chunk * kChunk + 4 * cl;
    const bool col_ok = col < sg.D;
    const float* V = sg.base + col;
    // Rows go through the registers in batches of kU per thread (rows rg, rg+kGroups, ...), two batches in
    // flight: the first two are requested BEFORE the softmax so their HBM latency hides behind it (a launch is
    // ~10 us: the fixed prologue matters as much as the streaming rate).  Row indices past the segment are
    // clamped to its last row and get weight 0, so EVERY batch — the ragged last one too — is kU independent
    // loads; single-row tail iterations would each expose one full memory latency (two or three per thread at
    // 283 / 101 rows).
    constexpr int kU = 4;
    float4 cur[kU], nxt[kU];
    // rows to stream: all of them, or only the rows before the valid length when the caller vouches that the
    // rest is zero padding (zero_past_len); the softmax below always spans all L scores
    const int n_stream = a.v.zero_past_len ? max(1, min(sg.L, sg.valid)) : sg.L;
    const int last_row = n_stream - 1, row_len = sg.D;
    auto fetch = [V, last_row, row_len](float4 (&dst)[kU], int first) {     // (captures by value: no struct reference)
#pragma unroll
        for (int u = 0; u < kU; ++u)
            dst[u] = *reinterpret_cast<const float4*>(V + (int64_t)min(first + u * kGroups, last_row) * row_len);
    };
    const bool stream = vec && col_ok;
    if (stream) {
        fetch(cur, rg);
        fetch(nxt, rg + kU * kGroups);
    }

    // ---- softmax of the score segment (every chunk of the segment recomputes it; L <= a few hundred).
    // w[] keeps the UNNORMALISED exp(s - max); the row sum only scales the final context, so the
    // value stream does not wait for it.  Two barriers before the stream, one after.
    const float* srow = a.scores + (int64_t)b * a.ld_s + sg.seg_off;
    float lmax = -INFINITY;
    for (int i = tid; i < sg.L; i += 256) {
        float s = srow[i];
        if (masked && i >= sg.valid) s = -INFINITY;
        w[i] = s;
        lmax = fmaxf(lmax, s);
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) sh[tid >> 6] = lmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float lsum = 0.f;
    for (int i = tid; i < sg.L; i += 256) {
        const float e = expf(w[i] - m);      // own element: written by this thread above
        w[i] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if ((tid & 63) == 0) sh[4 + (tid >> 6)] = lsum;
    __syncthreads();
    const float inv = 1.0f / (sh[4] + sh[5] + sh[6] + sh[7]);
    if (chunk == 0) {
        float* arow = a.attn + (int64_t)b * a.ld_a + sg.seg_off;
        for (int i = tid; i < sg.L; i += 256) arow[i] = w[i] * inv;
    }

    // ---- weighted row sum over this workgroup's kChunk columns
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
    if (col_ok) {
        if (vec) {
            for (int base = rg; base < n_stream; base += kU * kGroups) {
                float wv[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int i = base + u * kGroups;
                    wv[u] = i < n_stream ? w[i] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < kU; u += 2) {
                    acc0.x += wv[u] * cur[u].x; acc0.y += wv[u] * cur[u].y; acc0.z += wv[u] * cur[u].z; acc0.w += wv[u] * cur[u].w;
                    acc1.x += wv[u + 1] * cur[u + 1].x; acc1.y += wv[u + 1] * cur[u + 1].y;
                    acc1.z += wv[u + 1] * cur[u + 1].z; acc1.w += wv[u + 1] * cur[u + 1].w;
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) cur[u] = nxt[u];
                if (base + 2 * kU * kGroups < n_stream) fetch(nxt, base + 2 * kU * kGroups);
            }
        } else {
            for (int i = rg; i < n_stream; i += kGroups) {
                const float* r = V + (int64_t)i * sg.D;
                const float w0 = w[i];
                acc0.x += w0 * r[0];
                if (col + 1 < sg.D) acc0.y += w0 * r[1];
                if (col + 2 < sg.D) acc0.z += w0 * r[2];
                if (col + 3 < sg.D) acc0.w += w0 * r[3];
            }
        }
    }
    acc0.x += acc1.x; acc0.y += acc1.y; acc0.z += acc1.z; acc0.w += acc1.w;
    *reinterpret_cast<float4*>(&red[rg * kChunk + 4 * cl]) = acc0;
    __syncthreads();
    if (tid < kChunk) {
        const int c = chunk * kChunk + tid;
        if (c < sg.D) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < kGroups; ++r) s += red[r * kChunk + tid];
            a.ctx[(int64_t)b * a.ld_c + sg.ctx_off + c] = s * inv;
        }
    }
}

// Work items in dispatch order: every text item of the batch first (Lt rows each: the heavy ones), then the video
// items, then the audio items (Lav rows; audio in 32-column chunks).  Workgroups are handed to the CUs in launch order,
// so at config 2 every CU draws 2 text items (72 KB each), 2 video items (26 KB) and 1 audio item (13 KB): 209 KB,
// the same for all 256 CUs.  With the natural (question, modality, chunk) order some CUs drew 3 heavy + 2 light items,
// 27% above the mean, and every CU streams at about the same rate (MI355X_MICROARCH.md: ~10 B/clk/CU from HBM).
template <int kChunk>
__global__ __launch_bounds__(256) void attn_softmax_context_fwd_kernel(AttnFwdK a) {
    __shared__ float w[kMaxRows];
    __shared__ __attribute__((aligned(16))) float red[1024];       // (256 / (chunk/4)) row groups x chunk columns
    __shared__ float sh[8];
    int n = blockIdx.x;
    const int text_items = a.v.B * a.chunks_text, video_items = a.v.B * a.chunks_video;
    if (n < text_items) {
        const int b = n / a.chunks_text;
        attn_fwd_item<kChunk>(a, 0, b, n - b * a.chunks_text, w, red, sh);
    } else if (n - text_items < video_items) {
        n -= text_items;
        const int b = n / a.chunks_video;
        attn_fwd_item<kChunk>(a, 2, b, n - b * a.chunks_video, w, red, sh);
    } else {
        n -= text_items + video_items;
        const int b = n / a.chunks_audio;
        if (a.audio_chunk == 32) attn_fwd_item<32>(a, 1, b, n - b * a.chunks_audio, w, red, sh);
        else attn_fwd_item<kChunk>(a, 1, b, n - b * a.chunks_audio, w, red, sh);
    }
}

struct AttnBwdK {
    mmqg_attn_values v;
    const float* dctx; int ld_c;
    float* dscores; int ld_ds;
    int blocks_text, blocks_audio, blocks_video;
    int vec_text, vec_audio, vec_video;
    // fused softmax backward (both non-null): dscores = attn * (d(attn) - dot) with
    // dot = sum_j attn_j d(attn)_j = (sum_j attn_j V_j) . dctx = ctx . dctx — the forward's context, no row pass
    const float* attn; int ld_a;
    const float* ctx; int ld_x;
};

// d(attn)[b][seg+i] = V[b][i][:] . dctx[b][ctx_off : ctx_off + D]
// One workgroup per (question, modality, 32-row block), text blocks of the whole batch first (the heavy items,
// as in the forward kernel).  A wave takes 4 rows at a time — lanes across the contiguous row — so 4 x D/256
// independent 16-byte loads are in flight per lane before the first shuffle reduction (one row at a time exposed
// a full memory latency per row).
__global__ __launch_bounds__(256) void attn_dweights_kernel(AttnBwdK a) {
    int n = blockIdx.x, modality = 0, b, blk;
    const int text_items = a.v.B * a.blocks_text;
    if (n < text_items) {
        b = n / a.blocks_text; blk = n - b * a.blocks_text;
    } else {
        n -= text_items;
        const int per_q = a.blocks_audio + a.blocks_video;
        b = n / per_q; blk = n - b * per_q; modality = 1;
        if (blk >= a.blocks_audio) { blk -= a.blocks_audio; modality = 2; }
    }
    const Segment sg = pick_segment(a.v, modality, b);
    const bool vec = modality == 0 ? a.vec_text : (modality == 1 ? a.vec_audio : a.vec_video);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* g = a.dctx + (int64_t)b * a.ld_c + sg.ctx_off;
    float* out = a.dscores + (int64_t)b * a.ld_ds + sg.seg_off;
    const int row_end = min(sg.L, (blk + 1) * kRowBlock);
    const int n_stream = a.v.zero_past_len ? min(sg.L, sg.valid) : sg.L;      // rows past it are zero padding
    constexpr int kR = 4;
    const bool fused = a.attn != nullptr;
    // fused path: dot = ctx . dctx, accumulated inside the first batch's column loop (the ctx loads travel with
    // the first value rows instead of in front of them)
    float dot = 0.f;
    bool need_dot = fused;
    const float* cx = fused ? a.ctx + (int64_t)b * a.ld_x + sg.ctx_off : nullptr;
    const float* arow = fused ? a.attn + (int64_t)b * a.ld_a + sg.seg_off : nullptr;
    for (int i0 = blk * kRowBlock + wave * kR; i0 < row_end; i0 += 4 * kR) {
        float acc[kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) acc[r] = 0.f;
        // the rows' attention weights (fused path) are requested together with the value rows: lane r holds row i0 + r's
        float a_mine = 0.f;
        if (fused && lane < kR && i0 + lane < row_end) a_mine = arow[i0 + lane];
        if (i0 < n_stream) {
            const int last = n_stream - 1;
            if (vec) {
                for (int c = 4 * lane; c < sg.D; c += 256) {
                    const float4 y = *reinterpret_cast<const float4*>(g + c);
                    if (need_dot) {
                        const float4 z = *reinterpret_cast<const float4*>(cx + c);
                        dot += z.x * y.x + z.y * y.y + z.z * y.z + z.w * y.w;
                    }
                    float4 x[kR];
#pragma unroll
                    for (int r = 0; r < kR; ++r)      // rows past the block / the stream: re-read a valid row, result unused
                        x[r] = *reinterpret_cast<const float4*>(sg.base + (int64_t)min(i0 + r, last) * sg.D + c);
#pragma unroll
                    for (int r = 0; r < kR; ++r) acc[r] += x[r].x * y.x + x[r].y * y.y + x[r].z * y.z + x[r].w * y.w;
                }
            } else {
                for (int c = lane; c < sg.D; c += 64) {
                    const float y = g[c];
                    if (need_dot) dot += cx[c] * y;
#pragma unroll
                    for (int r = 0; r < kR; ++r) acc[r] += sg.base[(int64_t)min(i0 + r, last) * sg.D + c] * y;
                }
            }
        } else if (need_dot) {                      // nothing to stream for this wave: the dot still has to be formed
            for (int c = lane; c < sg.D; c += 64) dot += cx[c] * g[c];
        }
        if (need_dot) { dot = wave_sum(dot); need_dot = false; }
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const float v = wave_sum(acc[r]);
            const float ai = __shfl(a_mine, r, 64);
            const int i = i0 + r;
            if (lane == 0 && i < row_end) {
                const float da = i < n_stream ? v : 0.f;
                out[i] = fused ? ai * (da - dot) : da;
            }
        }
    }
}

// in place: ds[i] = a[i] * (da[i] - sum_j a[j] da[j]) per (question, segment)
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(mmqg_attn_values v, const float* attn, int ld_a,
                                                               float* dscores, int ld_ds, const float* dattn,
                                                               int ld_da) {
    __shared__ float sh[4];
    const int b = blockIdx.y, modality = blockIdx.x;
    const int L = modality == 0 ? v.Lt : v.Lav;
    const int off = modality == 0 ? 0 : (modality == 1 ? v.Lt : v.Lt + v.Lav);
    const float* ar = attn + (int64_t)b * ld_a + off;
    float* dr = dscores + (int64_t)b * ld_ds + off;
    const float* er = dattn ? dattn + (int64_t)b * ld_da + off : nullptr;   // gradient of the returned weights
    float part = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) {
        const float da = dr[i] + (er ? er[i] : 0.f);
        dr[i] = da;
        part += ar[i] * da;
    }
    const float dot = block_sum(part, sh);
    for (int i = threadIdx.x; i < L; i += 256) dr[i] = ar[i] * (dr[i] - dot);
}

struct DValK {
    int T, B, n_rows, D;
    const float* attn; int64_t attn_stride_t; int ld_a; int seg_off;
    const float* dctx; int64_t dctx_stride_t; int ld_c; int ctx_off;
    float* out; int64_t out_stride_row; int64_t out_stride_b;
    int accumulate;
};
constexpr int kDvRows = 8;
constexpr int kDvMaxT = 256;

template <int W>   // W = 4: float4 columns, W = 1: scalar columns
__global__ __launch_bounds__(256) void attn_dvalues_kernel(DValK a) {
    __shared__ float aw[kDvMaxT * kDvRows];
    const int b = blockIdx.y, row0 = blockIdx.x * kDvRows;
    for (int e = threadIdx.x; e < a.T * kDvRows; e += blockDim.x) {
        const int t = e / kDvRows, r = e % kDvRows;
        const int row = row0 + r;
        aw[e] = row < a.n_rows ? a.attn[t * a.attn_stride_t + (int64_t)b * a.ld_a + a.seg_off + row] : 0.f;
    }
    __syncthreads();
    const int ncol = a.D / W;
    for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
        float acc[kDvRows][W];
#pragma unroll
        for (int r = 0; r < kDvRows; ++r)
#pragma unroll
            for (int k = 0; k < W; ++k) acc[r][k] = 0.f;
        // eight tokens' context gradients in flight at a time (as a loop of one load per token every load was waited for
        // where it was issued: T dependent round trips for a kernel that is 12 us long at T = 20)
        constexpr int kTU = 8;
        const float* src0 = a.dctx + (int64_t)b * a.ld_c + a.ctx_off + c * W;
        for (int t0 = 0; t0 < a.T; t0 += kTU) {
            float x[kTU][W];
#pragma unroll
            for (int u = 0; u < kTU; ++u) {
                const float* src = src0 + (int64_t)min(t0 + u, a.T - 1) * a.dctx_stride_t;
                if constexpr (W == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z; x[u][3] = q.w;
                } else {
                    x[u][0] = src[0];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kTU; ++u) {
                const int t = t0 + u;
                if (t < a.T) {
#pragma unroll
                    for (int r = 0; r < kDvRows; ++r) {
                        const float wv = aw[t * kDvRows + r];
#pragma unroll
                        for (int k = 0; k < W; ++k) acc[r][k] += wv * x[u][k];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kDvRows; ++r) {
            const int row = row0 + r;
            if (row >= a.n_rows) continue;
            float* dst = a.out + (int64_t)b * a.out_stride_b + (int64_t)row * a.out_stride_row + c * W;
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = a.accumulate ? dst[k] + acc[r][k] : acc[r][k];
        }
    }
}

inline bool vec_ok(const float* p, int64_t stride_b, int D) {
    return mmqg::aligned16(p) && (stride_b % 4 == 0) && (D % 4 == 0);
}

int check_values(const mmqg_attn_values& v, const char* who) {
    MMQG_REQUIRE(v.B >= 0 && v.Lt > 0 && v.Lav > 0 && v.H > 0 && v.Da > 0 && v.Dv > 0, "%s: bad attention shape", who);
    MMQG_REQUIRE(v.text && v.audio && v.video, "%s: null value tensor", who);
    MMQG_REQUIRE(v.Lt <= kMaxRows && v.Lav <= kMaxRows, "%s: segment longer than %d rows", who, kMaxRows);
    MMQG_REQUIRE(v.mask_mode == MMQG_MASK_REFERENCE_NOOP || v.mask_mode == MMQG_MASK_INTENDED, "%s: bad mask_mode", who);
    MMQG_REQUIRE(v.mask_mode == MMQG_MASK_REFERENCE_NOOP || (v.text_len && v.av_len),
                 "%s: MMQG_MASK_INTENDED needs text_len and av_len", who);
    MMQG_REQUIRE(!v.zero_past_len || (v.text_len && v.av_len), "%s: zero_past_len needs text_len and av_len", who);
    return 0;
}

}  // namespace

namespace mmqg {

int attn_softmax_context_fwd(const mmqg_attn_values& v, const float* scores, int ld_s, float* attn, int ld_a,
                             float* ctx, int ld_c, hipStream_t s) {
    MMQG_TRY(check_values(v, "attn_softmax_context_fwd"));
    if (v.B == 0) return 0;
    const int S = v.Lt + 2 * v.Lav;
    MMQG_REQUIRE(scores && attn && ctx, "attn_softmax_context_fwd: null pointer");
    MMQG_REQUIRE(scores != attn, "attn_softmax_context_fwd: attn must not alias scores");
    MMQG_REQUIRE(ld_s >= S && ld_a >= S && ld_c >= v.H + v.Da + v.Dv, "attn_softmax_context_fwd: leading dimension too small");
    AttnFwdK k;
    k.v = v; k.scores = scores; k.ld_s = ld_s; k.attn = attn; k.ld_a = ld_a; k.ctx = ctx; k.ld_c = ld_c;
    static const int chunk = [] { const char* e = getenv("MMQG_ATTN_CHUNK"); return e ? atoi(e) : 64; }();
    k.chunks_text = ceil_div(v.H, chunk);
    k.audio_chunk = chunk == 64 ? 32 : chunk;
    k.chunks_audio = ceil_div(v.Da, k.audio_chunk);
    k.chunks_video = ceil_div(v.Dv, chunk);
    k.vec_text = vec_ok(v.text, v.text_stride_b, v.H);
    k.vec_audio = vec_ok(v.audio, v.audio_stride_b, v.Da);
    k.vec_video = vec_ok(v.video, v.video_stride_b, v.Dv);
    dim3 grid((unsigned)((k.chunks_text + k.chunks_audio + k.chunks_video) * v.B));
    if (chunk == 64) hipLaunchKernelGGL(attn_softmax_context_fwd_kernel<64>, grid, dim3(256), 0, s, k);
    else if (chunk == 32) hipLaunchKernelGGL(attn_softmax_context_fwd_kernel<32>, grid, dim3(256), 0, s, k);
    else if (chunk == 256) hipLaunchKernelGGL(attn_softmax_context_fwd_kernel<256>, grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL(attn_softmax_context_fwd_kernel<128>, grid, dim3(256), 0, s, k);
    return check_launch("attn_softmax_context_fwd");
}

static int attn_context_bwd_impl(const mmqg_attn_values& v, const float* attn, int ld_a, const float* ctx, int ld_x,
                                 const float* dctx, int ld_c, const float* dattn, int ld_da, float* dscores, int ld_ds,
                                 hipStream_t s);

int attn_context_bwd(const mmqg_attn_values& v, const float* attn, int ld_a, const float* dctx, int ld_c,
                     const float* dattn, int ld_da, float* dscores, int ld_ds, hipStream_t s) {
    return attn_context_bwd_impl(v, attn, ld_a, nullptr, 0, dctx, ld_c, dattn, ld_da, dscores, ld_ds, s);
}

// the same with the forward's contexts at hand: ONE kernel (no separate softmax-Jacobian pass)
int attn_context_bwd_fused(const mmqg_attn_values& v, const float* attn, int ld_a, const float* ctx, int ld_x,
                           const float* dctx, int ld_c, float* dscores, int ld_ds, hipStream_t s) {
    MMQG_REQUIRE(ctx && ld_x >= v.H + v.Da + v.Dv, "attn_context_bwd_fused: bad ctx");
    return attn_context_bwd_impl(v, attn, ld_a, ctx, ld_x, dctx, ld_c, nullptr, 0, dscores, ld_ds, s);
}

static int attn_context_bwd_impl(const mmqg_attn_values& v, const float* attn, int ld_a, const float* ctx, int ld_x,
                                 const float* dctx, int ld_c, const float* dattn, int ld_da, float* dscores, int ld_ds,
                                 hipStream_t s) {
    MMQG_TRY(check_values(v, "attn_context_bwd"));
    if (v.B == 0) return 0;
    const int S = v.Lt + 2 * v.Lav;
    MMQG_REQUIRE(attn && dctx && dscores, "attn_context_bwd: null pointer");
    MMQG_REQUIRE(ld_a >= S && ld_ds >= S && ld_c >= v.H + v.Da + v.Dv, "attn_context_bwd: leading dimension too small");
    MMQG_REQUIRE(!dattn || ld_da >= S, "attn_context_bwd: ld_da too small");
    AttnBwdK k;
    k.v = v; k.dctx = dctx; k.ld_c = ld_c; k.dscores = dscores; k.ld_ds = ld_ds;
    k.blocks_text = ceil_div(v.Lt, kRowBlock);
    k.blocks_audio = ceil_div(v.Lav, kRowBlock);
    k.blocks_video = ceil_div(v.Lav, kRowBlock);
    // the fused form also reads the saved contexts with 16-byte loads: they are an operand of their own (public
    // entry point: ctx / ld_x are independent of dctx / ld_c)
    const bool g_ok = aligned16(dctx) && (ld_c % 4 == 0) && (v.H % 4 == 0) && (v.Da % 4 == 0) &&
                      (!ctx || (aligned16(ctx) && ld_x % 4 == 0));
    k.vec_text = g_ok && vec_ok(v.text, v.text_stride_b, v.H);
    k.vec_audio = g_ok && vec_ok(v.audio, v.audio_stride_b, v.Da);
    k.vec_video = g_ok && vec_ok(v.video, v.video_stride_b, v.Dv);
    dim3 grid((unsigned)((k.blocks_text + k.blocks_audio + k.blocks_video) * v.B));
    const bool fused = ctx != nullptr && dattn == nullptr;
    k.attn = fused ? attn : nullptr; k.ld_a = ld_a; k.ctx = fused ? ctx : nullptr; k.ld_x = ld_x;
    hipLaunchKernelGGL(attn_dweights_kernel, grid, dim3(256), 0, s, k);
    MMQG_TRY(check_launch("attn_dweights"));
    if (fused) return 0;
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3(3, v.B), dim3(256), 0, s, v, attn, ld_a, dscores, ld_ds, dattn, ld_da);
    return check_launch("attn_softmax_bwd");
}

int attn_dvalues(int T, int B, int n_rows, int D, const float* attn, int64_t attn_stride_t, int ld_a, int seg_off,
                 const float* dctx, int64_t dctx_stride_t, int ld_c, int ctx_off, float* out, int64_t out_stride_row,
                 int64_t out_stride_b, int accumulate, hipStream_t s) {
    MMQG_REQUIRE(T >= 0 && B >= 0 && n_rows >= 0 && D > 0, "attn_dvalues: bad shape");
    if (B == 0 || n_rows == 0) return 0;
    MMQG_REQUIRE(T <= kDvMaxT, "attn_dvalues: more than %d steps", kDvMaxT);
    MMQG_REQUIRE(attn && dctx && out, "attn_dvalues: null pointer");
    DValK k{T, B, n_rows, D, attn, attn_stride_t, ld_a, seg_off, dctx, dctx_stride_t, ld_c, ctx_off,
            out, out_stride_row, out_stride_b, accumulate};
    dim3 grid(ceil_div(n_rows, kDvRows), B);
    const bool vec = (D % 4 == 0) && aligned16(dctx) && (ld_c % 4 == 0) && (ctx_off % 4 == 0) &&
                     (dctx_stride_t % 4 == 0) && aligned16(out) && (out_stride_row % 4 == 0) && (out_stride_b % 4 == 0);
    if (vec) {
        const int threads = std::min(256, ceil_div(D / 4, 64) * 64);
        hipLaunchKernelGGL(attn_dvalues_kernel<4>, grid, dim3(threads), 0, s, k);
    } else {
        const int threads = std::min(256, ceil_div(D, 64) * 64);
        hipLaunchKernelGGL(attn_dvalues_kernel<1>, grid, dim3(threads), 0, s, k);
    }
    return check_launch("attn_dvalues");
}

}  // namespace mmqg
