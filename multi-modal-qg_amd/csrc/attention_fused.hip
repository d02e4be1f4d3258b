// Fused decoder-attention forward for the training time loop: score product + softmax + context in ONE launch
// (model/decoder.py:78-95: s = Linear([emb | h_top]) -> softmax -> bmm, three modalities).  The embedded-word half of
// the scores (+ bias) is hoisted over all steps (sequence.hip); what is left per step is
// scores += h_top(t-1) W_attn[:, E:]^T — until round 3 a launch of its own on the dependent chain (15.7 us for a
// 64 x 485 x 512 product) in front of the attention launch (11.1 us).
//
// Here a workgroup owns (question b, modality, a RANGE OF ROWS of the segment) instead of a column chunk, so it needs
// only its own rows' scores and forms them itself: one wave per row, lanes along the query (whole 1-2 KB runs of the
// score matrix, which the batch shares and the L2s keep).  It takes a LOCAL softmax over its rows (max m_i,
// unnormalised weights e_r = exp(s_r - m_i), sum l_i), streams its value rows exactly once — whole contiguous rows,
// 16 bytes per lane, eight loads in flight per thread, the first ones requested before the scores — and writes a
// partial record {m_i, l_i, c_i = sum_r e_r V_r}.  The parts of a (question, modality) meet at a ticket: the
// workgroup whose atomic add comes last merges them flash-attention style (M = max m_i,
// ctx = sum_i c_i e^(m_i - M) / sum_i l_i e^(m_i - M)), normalises the attention weights and resets the ticket.
//
// Hand-off without fences (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md valid-forms table, first row):
// every partial byte is stored write-through (sc1), every storing wave drains its stores (s_waitcnt vmcnt(0)) before
// the workgroup barrier, ONE lane then adds to the ticket with a returning agent-scope atomic; the last arriver loads
// the partials with sc1 loads after its add has returned, its other waves after a workgroup barrier it then joins.
//
// HBM-bound like the unfused kernel: per (question, step) 4 (Lt H + Lav Da + Lav Dv) value bytes; the score matrix
// ((Lt + 2 Lav) x H floats = 1 MB at config.py widths) counts once per launch.
#include <stdlib.h>

#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

using namespace mmqg;

typedef float fx4 __attribute__((ext_vector_type(4)));
typedef unsigned ux4 __attribute__((ext_vector_type(4)));

constexpr int kMaxPartRows = 128;    // rows of one part: one wave takes the part's softmax, two rows per lane
constexpr int kMaxParts = 32;        // parts of one (question, modality)
constexpr int kMaxQ4 = 4;            // float4 per lane of a score-matrix row: Hq <= 1024

struct FusedK {
    mmqg_attn_values v;
    const float* pre; int ld_s;          // [B][ld_s] hoisted half of the scores (+ bias)
    const float* h; int ld_h;            // [B][Hq] the query's recurrent half (h_top of the previous step)
    const float* W; int ld_w; int Hq;    // score matrix rows s = 0 .. S-1: W[s * ld_w .. + Hq)
    float* attn; int ld_a;
    float* ctx; int ld_c;
    float* part;                         // [B][P0 + P1 + P2][pstride]: {m, l, -, -, c[D]}
    unsigned* ticket;                    // [B][3], zero between launches
    int R0, R1, R2, P0, P1, P2;          // rows per part / parts per modality (text, audio, video)
    int pstride;
    int64_t part_bytes;
};

struct Seg {
    const float* base; int L, D, seg_off, ctx_off, valid, R, P, pfirst, modality;
};

__device__ __forceinline__ Seg pick(const FusedK& a, int modality, int b) {
    const mmqg_attn_values& v = a.v;
    Seg s;
    s.modality = modality;
    if (modality == 0) {
        s.base = v.text + (int64_t)b * v.text_stride_b; s.L = v.Lt; s.D = v.H; s.seg_off = 0; s.ctx_off = 0;
        s.valid = v.text_len ? v.text_len[b] : v.Lt; s.R = a.R0; s.P = a.P0; s.pfirst = 0;
    } else if (modality == 1) {
        s.base = v.audio + (int64_t)b * v.audio_stride_b; s.L = v.Lav; s.D = v.Da; s.seg_off = v.Lt; s.ctx_off = v.H;
        s.valid = v.av_len ? v.av_len[b] : v.Lav; s.R = a.R1; s.P = a.P1; s.pfirst = a.P0;
    } else {
        s.base = v.video + (int64_t)b * v.video_stride_b; s.L = v.Lav; s.D = v.Dv; s.seg_off = v.Lt + v.Lav;
        s.ctx_off = v.H + v.Da; s.valid = v.av_len ? v.av_len[b] : v.Lav; s.R = a.R2; s.P = a.P2; s.pfirst = a.P0 + a.P1;
    }
    return s;
}

__device__ __forceinline__ float ld_rlx(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // global_load_dword sc1
}
__device__ __forceinline__ void st_rlx(float* p, float x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);              // global_store_dword sc1
}

// Work items in dispatch order: question-major, the P0 + P2 + P1 parts of a question next to each other (text, video,
// audio).  The plan makes the parts per question a multiple of 8, and workgroups are dealt round-robin over the 8 XCDs
// (observed, speed only): part pp of EVERY question then lands on XCD pp % 8, so an XCD's L2 keeps just the score-matrix
// rows of its own parts — in the step that matrix is not L2-resident when the launch starts (the layer-step kernels in
// between stream 25 MB of LSTM weights through the L2s), and 64 questions re-read each row.
// NQ = float4 per lane of a score-matrix row (Hq <= 256 NQ).  <= 128 VGPRs: four workgroups per CU.
template <int NQ>
__global__ __launch_bounds__(256, 4) void attn_fused_fwd_kernel(FusedK a) {
    constexpr int kRowBatch = NQ == 4 ? 4 : 8;     // score rows a wave has in flight (a part usually has <= 8 rows per wave)
    __shared__ float sc[kMaxPartRows];                                  // the part's scores, then its weights e_r
    __shared__ __attribute__((aligned(16))) float red[1024];            // row-group partials: one float4 per streaming thread
    __shared__ float pm[kMaxParts], pscale[kMaxParts];
    __shared__ unsigned last_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Pq = a.P0 + a.P1 + a.P2;
    const int b = blockIdx.x / Pq;
    int p = blockIdx.x - b * Pq, modality;
    if (p < a.P0) modality = 0;
    else if (p < a.P0 + a.P2) { modality = 2; p -= a.P0; }
    else { modality = 1; p -= a.P0 + a.P2; }
    const Seg sg = pick(a, modality, b);
    const int r0 = p * sg.R;
    const int R = min(sg.R, sg.L - r0);                                 // rows of this part (>= 1 by construction)
    const bool masked = a.v.mask_mode == MMQG_MASK_INTENDED;
    // rows whose values are streamed: all, or only those before the valid length when the caller vouches for zeros
    const int n_stream = a.v.zero_past_len ? max(0, min(R, sg.valid - r0)) : R;

    // ---- value stream mapping: kLanes float4 column lanes x kGroups row groups (D % 4 == 0, D <= 1024)
    const int kLanes = sg.D >> 2;
    const int kGroups = min(256 / kLanes, 8);
    const int cl = tid % kLanes, rg = tid / kLanes;
    const bool streamer = rg < kGroups;
    const float* V = sg.base + (int64_t)r0 * sg.D + 4 * cl;
    constexpr int kU = 2;               // 2 x 2 loads in flight per thread: four workgroups per CU keep 64 KB in flight
    fx4 cur[kU], nxt[kU];
    const int last_row = max(n_stream - 1, 0);
    auto fetch = [V, last_row, kGroups, D = sg.D](fx4 (&dst)[kU], int first) {
#pragma unroll
        for (int u = 0; u < kU; ++u) dst[u] = *reinterpret_cast<const fx4*>(V + (int64_t)min(first + u * kGroups, last_row) * D);
    };
    const bool stream = streamer && n_stream > 0;
    if (stream) {
        fetch(cur, rg);
        fetch(nxt, rg + kU * kGroups);
    }

    // ---- scores of the part's rows: wave w takes rows w, w + 4, ...; lanes along the query
    {
        fx4 hq[NQ];
        const float* hrow = a.h + (int64_t)b * a.ld_h;
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
            const int col = 4 * lane + 256 * c;
            hq[c] = col < a.Hq ? *reinterpret_cast<const fx4*>(hrow + col) : fx4{0.f, 0.f, 0.f, 0.f};
        }
        // lane k of wave w keeps the hoisted half (and, in the end, the score) of row w + 4 k: one load per lane up
        // front instead of a dependent scalar load per row
        const float* prow = a.pre + (int64_t)b * a.ld_s + sg.seg_off + r0;
        const int my_row = wave + 4 * lane;
        float my_s = my_row < R ? prow[my_row] : 0.f;
        const float* Wseg = a.W + (int64_t)(sg.seg_off + r0) * a.ld_w;
        for (int i0 = wave; i0 < R; i0 += 4 * kRowBatch) {
            fx4 wv[kRowBatch][NQ];
#pragma unroll
            for (int k = 0; k < kRowBatch; ++k) {
                const int i = min(i0 + 4 * k, R - 1);
#pragma unroll
                for (int c = 0; c < NQ; ++c) {
                    const int col = 4 * lane + 256 * c;
                    wv[k][c] = col < a.Hq ? *reinterpret_cast<const fx4*>(Wseg + (int64_t)i * a.ld_w + col) : fx4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int k = 0; k < kRowBatch; ++k) {
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < NQ; ++c)
                    d += wv[k][c].x * hq[c].x + wv[k][c].y * hq[c].y + wv[k][c].z * hq[c].z + wv[k][c].w * hq[c].w;
                d = wave_sum(d);
                if (lane == (i0 >> 2) + k) my_s += d;               // row i0 + 4 k = wave + 4 lane for this lane
            }
        }
        if (my_row < R) {
            if (masked && r0 + my_row >= sg.valid) my_s = -INFINITY;
            sc[my_row] = my_s;
        }
    }
    __syncthreads();
    // ---- local softmax of the part (R <= 128: every wave does it for itself, two rows per lane)
    const float s_a = lane < R ? sc[lane] : -INFINITY, s_b = lane + 64 < R ? sc[lane + 64] : -INFINITY;
    const float m = wave_max(fmaxf(s_a, s_b));
    const float e_a = s_a != -INFINITY ? expf(s_a - m) : 0.f, e_b = s_b != -INFINITY ? expf(s_b - m) : 0.f;
    const float l = wave_sum(e_a + e_b);
    __syncthreads();                                                    // everyone has read sc[] as scores
    float* arow = a.attn + (int64_t)b * a.ld_a + sg.seg_off + r0;
    if (wave == 0) {                                                    // unnormalised weights; the last arriver rescales
        if (lane < R) { sc[lane] = e_a; st_rlx(arow + lane, e_a); }
        if (lane + 64 < R) { sc[lane + 64] = e_b; st_rlx(arow + lane + 64, e_b); }
    }
    __syncthreads();

    // ---- weighted row sum over this part's rows
    fx4 acc0 = fx4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (stream) {
        for (int base = rg; base < n_stream; base += kU * kGroups) {
            float wv[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = base + u * kGroups;
                wv[u] = i < n_stream ? sc[i] : 0.f;
            }
            acc0 += wv[0] * cur[0];
            acc1 += wv[1] * cur[1];
#pragma unroll
            for (int u = 0; u < kU; ++u) cur[u] = nxt[u];
            if (base + 2 * kU * kGroups < n_stream) fetch(nxt, base + 2 * kU * kGroups);
        }
    }
    acc0 += acc1;
    // combine the row groups through LDS (at most 8 groups x 128 float4 lanes, or 1-2 groups of up to 256 lanes)
    if (streamer) *reinterpret_cast<fx4*>(&red[(rg * kLanes + cl) * 4]) = acc0;
    __syncthreads();
    float* prec = a.part + ((int64_t)b * (a.P0 + a.P1 + a.P2) + sg.pfirst + p) * a.pstride;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, (int)a.part_bytes, 0x00020000);
    const int prec_off = (int)((prec - a.part) * 4);
    if (tid < kLanes) {
        fx4 s = *reinterpret_cast<const fx4*>(&red[tid * 4]);
        for (int g = 1; g < kGroups; ++g) s += *reinterpret_cast<const fx4*>(&red[(g * kLanes + tid) * 4]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ux4, s), rs, prec_off + 16 + 16 * tid, 0, 16);
    }
    if (tid == 0) { st_rlx(prec, m); st_rlx(prec + 1, l); }
    // ---- ticket: every storing wave drains its write-through stores, then ONE lane adds (returning atomic)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* tk = a.ticket + b * 3 + modality;
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (old == (unsigned)sg.P - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!last_flag) return;

    // ---- last arriver of (b, modality): merge the parts
    const float* pbase = a.part + ((int64_t)b * (a.P0 + a.P1 + a.P2) + sg.pfirst) * a.pstride;
    if (tid < 64) {
        float mi = -INFINITY, li = 0.f;
        if (tid < sg.P) { mi = ld_rlx(pbase + (int64_t)tid * a.pstride); li = ld_rlx(pbase + (int64_t)tid * a.pstride + 1); }
        const float M = wave_max(mi);
        const float sci = (tid < sg.P && mi != -INFINITY) ? expf(mi - M) : 0.f;     // a fully masked part weighs nothing
        const float Lsum = wave_sum(li * sci);
        if (tid < sg.P) { pm[tid] = sci / Lsum; }
        if (tid == 0) pscale[0] = M;      // (kept for debugging; M == -inf with L == 0 gives NaN like the reference's softmax)
    }
    __syncthreads();
    // contexts: thread (column lane cl4, half hf) sums the parts hf, hf + 2, ... — eight partial vectors requested
    // before the first is used (a loop of dependent write-through loads would cost a memory round trip per part)
    const int pfirst_off = (int)((pbase - a.part) * 4);
    {
        const int nh = kLanes <= 128 ? 2 : 1;                         // thread groups that split the parts between them
        const int cl4 = tid % kLanes, hf = tid / kLanes;
        fx4 sum = fx4{0.f, 0.f, 0.f, 0.f};
        if (hf < nh) {
            for (int i0 = hf; i0 < sg.P; i0 += 8 * nh) {
                fx4 c[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = min(i0 + nh * k, sg.P - 1);
                    c[k] = __builtin_bit_cast(fx4, __builtin_amdgcn_raw_buffer_load_b128(rs, pfirst_off + i * a.pstride * 4 + 16 + 16 * cl4, 0, 16));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + nh * k;
                    sum += (i < sg.P ? pm[min(i, sg.P - 1)] : 0.f) * c[k];
                }
            }
        }
        if (nh == 2 && hf == 1) *reinterpret_cast<fx4*>(&red[cl4 * 4]) = sum;
        __syncthreads();
        if (hf == 0) {
            if (nh == 2) sum += *reinterpret_cast<const fx4*>(&red[cl4 * 4]);
            *reinterpret_cast<fx4*>(a.ctx + (int64_t)b * a.ld_c + sg.ctx_off + 4 * cl4) = sum;
        }
    }
    float* afull = a.attn + (int64_t)b * a.ld_a + sg.seg_off;
    for (int r = tid; r < sg.L; r += 256) afull[r] = ld_rlx(afull + r) * pm[r / sg.R];
    if (tid == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
}

struct Plan { int R[3], P[3], pstride; int64_t part_bytes, ticket_off, total; };

// Parts of about equal cost — a row costs its value bytes (4 D) plus its score-matrix row (4 Hq) — about four
// workgroups per CU in the launch, and the parts per question nudged to a multiple of 8 (XCD locality, see the kernel).
bool make_plan(const mmqg_attn_values& v, int Hq, Plan& pl) {
    const int D[3] = {v.H, v.Da, v.Dv}, Lr[3] = {v.Lt, v.Lav, v.Lav};
    int64_t cost[3], total = 0;
    int Dmax = 0;
    for (int i = 0; i < 3; ++i) {
        if (D[i] % 4 != 0 || D[i] > 1024 || D[i] < 4 || Lr[i] < 1) return false;
        cost[i] = (int64_t)Lr[i] * 4 * (D[i] + Hq);
        total += cost[i];
        Dmax = std::max(Dmax, D[i]);
    }
    int Pq = std::max(8, std::min(64, (1024 / std::max(v.B, 1) + 7) / 8 * 8));
    Pq = std::min(Pq, std::max(8, (v.Lt + 2 * v.Lav) / 4 / 8 * 8));           // short segments: no one-row parts
    auto fit = [&](int i, int P, int& R, int& Pout) {
        P = std::max(1, std::min(P, std::min(Lr[i], kMaxParts)));
        P = std::max(P, ceil_div(Lr[i], kMaxPartRows));
        R = ceil_div(Lr[i], P);
        Pout = ceil_div(Lr[i], R);
    };
    int R[3], P[3];
    for (int i = 0; i < 3; ++i) fit(i, (int)std::max<int64_t>(1, (cost[i] * Pq + total / 2) / total), R[i], P[i]);
    for (int it = 0; it < 24 && (P[0] + P[1] + P[2]) % 8 != 0; ++it) {
        const int i = it % 3 == 0 ? 1 : (it % 3 == 1 ? 2 : 0);
        for (int cand = P[i] + 1; cand < P[i] + 5; ++cand) {
            int r, pp;
            fit(i, cand, r, pp);
            if (pp > P[i]) { R[i] = r; P[i] = pp; break; }
        }
    }
    for (int i = 0; i < 3; ++i) {
        if (R[i] > kMaxPartRows || P[i] > kMaxParts) return false;
        pl.R[i] = R[i]; pl.P[i] = P[i];
    }
    pl.pstride = 4 + Dmax;
    pl.part_bytes = (int64_t)v.B * (pl.P[0] + pl.P[1] + pl.P[2]) * pl.pstride * 4;
    if (pl.part_bytes >= (int64_t)1 << 31) return false;
    pl.ticket_off = (pl.part_bytes + 255) / 256 * 256;
    pl.total = pl.ticket_off + (int64_t)v.B * 3 * 4;
    return true;
}

}  // namespace

namespace mmqg {

// workspace bytes of the fused forward for these extents (0: shape not taken); the ticket words at its end must be
// zero before the first launch (the kernel leaves them zero)
int64_t attn_fused_ws_bytes(const mmqg_attn_values& v, int Hq) {
    if (v.B <= 0 || Hq % 4 != 0 || Hq > 256 * kMaxQ4 || Hq < 4) return 0;
    Plan pl;
    if (!make_plan(v, Hq, pl)) return 0;
    return pl.total;
}

// 0 = launched, 1 = not taken (operands not 16-byte aligned, query too wide, ...): the caller runs the score product
// and attn_softmax_context_fwd instead; < 0 = error
int attn_fused_fwd(const mmqg_attn_values& v, const float* pre, int ld_s, const float* h, int ld_h, const float* W, int ld_w,
                   int Hq, float* attn, int ld_a, float* ctx, int ld_c, float* ws, int64_t ws_bytes, hipStream_t s) {
    if (!ws || v.B <= 0) return 1;
    Plan pl;
    if (attn_fused_ws_bytes(v, Hq) == 0 || !make_plan(v, Hq, pl) || ws_bytes < pl.total) return 1;
    if (!aligned16(h) || ld_h % 4 || !aligned16(W) || ld_w % 4 || !aligned16(ctx) || ld_c % 4 || !aligned16(ws)) return 1;
    if (!aligned16(v.text) || v.text_stride_b % 4 || !aligned16(v.audio) || v.audio_stride_b % 4 || !aligned16(v.video) ||
        v.video_stride_b % 4 || v.H % 4 || v.Da % 4)
        return 1;
    const int S = v.Lt + 2 * v.Lav;
    MMQG_REQUIRE(pre && attn && ctx, "attn_fused_fwd: null pointer");
    MMQG_REQUIRE(ld_s >= S && ld_a >= S && ld_c >= v.H + v.Da + v.Dv && ld_w >= Hq && ld_h >= Hq, "attn_fused_fwd: leading dimension too small");
    MMQG_REQUIRE(v.mask_mode == MMQG_MASK_REFERENCE_NOOP || (v.text_len && v.av_len), "attn_fused_fwd: MMQG_MASK_INTENDED needs lengths");
    MMQG_REQUIRE(!v.zero_past_len || (v.text_len && v.av_len), "attn_fused_fwd: zero_past_len needs text_len and av_len");
    FusedK k;
    k.v = v; k.pre = pre; k.ld_s = ld_s; k.h = h; k.ld_h = ld_h; k.W = W; k.ld_w = ld_w; k.Hq = Hq;
    k.attn = attn; k.ld_a = ld_a; k.ctx = ctx; k.ld_c = ld_c;
    k.part = ws; k.ticket = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + pl.ticket_off);
    k.R0 = pl.R[0]; k.R1 = pl.R[1]; k.R2 = pl.R[2]; k.P0 = pl.P[0]; k.P1 = pl.P[1]; k.P2 = pl.P[2];
    k.pstride = pl.pstride; k.part_bytes = pl.part_bytes;
    const unsigned grid = (unsigned)(v.B * (pl.P[0] + pl.P[1] + pl.P[2]));
    if (Hq <= 256) hipLaunchKernelGGL(attn_fused_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, k);
    else if (Hq <= 512) hipLaunchKernelGGL(attn_fused_fwd_kernel<2>, dim3(grid), dim3(256), 0, s, k);
    else hipLaunchKernelGGL(attn_fused_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, k);
    return check_launch("attn_fused_fwd");
}

}  // namespace mmqg
