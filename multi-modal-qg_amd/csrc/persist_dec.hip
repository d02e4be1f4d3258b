// Persistent FORWARD time loop of the attention decoder (AttnDecoder.forward, model/decoder.py:74-107, driven one token
// at a time by train.py:171-175): ONE launch runs all T teacher-forced steps.  A step is a strict chain — the scores
// need h_top(t-1), the contexts need the scores, layer 0 needs the contexts, layer 1 needs layer 0, layer 2 needs layer
// 1 — so there is no wavefront over (layer, time) as in the text encoder: the launch has FIVE phases per token, each
// closed by the fence-free device-wide barrier of grid_barrier.h:
//   S    scores(t) = hoisted half + h_top(t-1) W_attn[:, E:]^T      tiles of 16 score columns, each on up to 4 workgroups
//                                                                    (16 questions each), K split over the 8 waves
//   ATT  three softmaxes + three contexts (the 54 MB value stream)  every CU: waves 1-7, one (question, modality, 64-column
//                                                                    chunk) item per wave or per 2 / 4 waves (rows split,
//                                                                    partials combined through LDS); the first 16 KB of a
//                                                                    wave's value rows are issued while the S barrier turns
//   L0   gates_0 = hoisted half + [ctx(t) | h_0(t-1)] [W_ih0c | W_hh0]^T -> cell     workgroups 0 .. H/4-1
//   L1   [h_0(t) | h_1(t-1)] -> cell                                                  workgroups H/4 .. H/2-1
//   L2   [h_1(t) | h_2(t-1)] -> cell                                                  the same workgroups
// The RECURRENT half of every layer product (h_l(t-1) W_hh_l^T) is formed ahead, in a window in which the workgroup would
// only wait (layer 1's while layer 0 runs, layer 2's behind the attention barrier, layer 0's — for the next token —
// while layer 1 runs), and kept in registers: a layer phase carries only the operand that really is late.
// What the launches pay per token and this kernel does not: the recurrent weights (31 MB at config 2) stay in LDS for
// the whole sequence — the attention's 54 MB per token flush the 32 MB of L2, so every layer-step LAUNCH re-fetches its
// 8-13 MB of weights from the Infinity Cache — and the argument fetch / drain of five launches.  What it pays instead:
// five barriers per token (2.5 us each + store acknowledgement + one load round) and only H/4 = 128 of the 256 CUs
// working in a layer phase (an output-stationary unit of 16 gate columns x K <= 1664 is 64-106 KB: one or two per CU).
// Exchange data (h of every layer, its dropped copy, the contexts, the scores) has ONE SLOT PER TOKEN: it is stored
// write-through (sc1) once and only read after the barrier behind its production, so plain cached loads cannot see an
// older copy.  c and h of a unit's rows live in registers for the whole sequence.  Measurements: DESIGN.md sections 4, 8.
#include <stdlib.h>

#include <algorithm>

#include "grid_barrier.h"
#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

using namespace mmqg;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kRows = 64;
constexpr int kRingAhead = 16;           // operand chunks (1 KB per wave) in flight: products in idle windows
constexpr int kRingLate = 16;             // ... products on the chain (layer 0: 12)
constexpr int kLdsBudget = 160 * 1024 - 1024;
constexpr int kMaxSeg = 320;             // longest score segment an attention wave keeps in LDS (Lt = 283)

struct DecArgs {
    int T, B, H, E, Cw, S, ldS, G;
    mmqg_attn_values v;
    const float* pre_scores;             // [T][B][ldS] hoisted half of the scores (+ bias)
    const float* w_attn_h; int ld_wa;    // score matrix rows s: w_attn_h + s * ld_wa (H floats): W_attn[:, E:]
    const float* w_ih0c; int ld_w0;      // [4H] rows, Cw floats each: W_ih0[:, E:]
    const float* w_hh0;                  // [4H][H]
    const float* w_ih1; const float* w_hh1; const float* w_ih2; const float* w_hh2;
    const float* b_ih1; const float* b_hh1; const float* b_ih2; const float* b_hh2;
    const float* h0; const float* c0; int64_t h0_stride_l;
    const int32_t* lens;
    float* gates;                        // [3][T][B][4H]; layer 0's slots hold the hoisted pre-activations on entry
    float* hs; float* cs;                // [3][T+1][B][H]
    float* hdrop;                        // [2][T][B][H] or null
    float* attn; float* ctx;             // [T][B][ldS], [T][B][Cw]
    float drop_p; int drop; uint64_t seed, stream_base; const int32_t* seed_off;
    // workspace: one buffer descriptor over hx | xd | cx | sx
    float* hx;                           // [3][T+1][H/4][64][4]   h_l(t) in slot t + 1
    int xd_off, cx_off, sx_off, ex_bytes;   // byte offsets of xd [2][T][H/4][64][4], cx [T][Cw/4][64][4], sx [T][64][ldS]
    gb::XBar* bar;
    unsigned* scnt;                      // [4] arrival counters of the score tiles, one per block of 16 questions, each on a 128-byte line of its own (null: a device-wide barrier behind the score phase)
    float* poison; unsigned* sticky_fail; unsigned* host_fail; unsigned expect_wg, max_spins;
    unsigned long long* trace;
};

__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const int32_t* off) {
    return off ? seed + (uint64_t)(uint32_t)off[0] * 0x9E3779B97F4A7C15ull : seed;
}

template <typename Rsrc>
__device__ __forceinline__ f32x4 ldx(const Rsrc& rs, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
template <typename Rsrc>
__device__ __forceinline__ void stx(const Rsrc& rs, int off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 16);
}

__device__ __forceinline__ void mfma_chunk(f32x4& acc0, f32x4& acc1, const f32x4& wt, const f32x4& x) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.x, x.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.y, x.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.z, x.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.w, x.w, acc1, 0, 0, 0);
}

// One wave's share of a product: n chunks of 16 k each for ONE block of 16 rows.  Chunk i's operand fragment sits at byte
// offset off + i * (4 * 64 * 16) of the exchange buffer (off includes the lane's share), its weight fragment at
// lds[wbase + i * 64 + lane].  The count is padded to a multiple of the ring depth R with the all-zero weight chunk at
// LDS float4 index 0, so the loops are branch-free (hipcc then counts the outstanding loads instead of draining them).
struct WaveProd { int n, off, wbase; };

__device__ __forceinline__ WaveProd make_prod(int n_total, int part, int nparts, int off0, int wbase0, int lane_off) {
    const int per = (n_total + nparts - 1) / nparts;
    const int lo = min(n_total, part * per);
    WaveProd w;
    w.n = max(1, min(n_total, lo + per) - lo);
    w.off = off0 + lo * (4 * kRows * 16) + lane_off;
    w.wbase = wbase0 + lo * 64;
    return w;
}

// (not inlined: seven call sites; inlined, hipcc runs the kernel out of registers — 54 spilled — at R = 16)
template <int R, typename Rsrc>
__device__ __attribute__((noinline)) f32x4 wave_product(const Rsrc rs, const WaveProd w, const f32x4* lds, int lane) {
    constexpr int cb = 4 * kRows * 16;
    const int n_pad = (w.n + R - 1) / R * R;
    f32x4 ring[R];
#pragma unroll
    for (int d = 0; d < R; ++d) ring[d] = ldx(rs, w.off + min(d, w.n - 1) * cb);
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    f32x4 wcur = lds[w.wbase + lane];
    for (int i0 = 0; i0 + R < n_pad; i0 += R) {
#pragma unroll
        for (int d = 0; d < R; ++d) {
            const int in = i0 + d + 1;
            const f32x4 wnext = lds[(in < w.n ? w.wbase + in * 64 : 0) + lane];
            mfma_chunk(acc0, acc1, wcur, ring[d]);
            ring[d] = ldx(rs, w.off + min(i0 + R + d, w.n - 1) * cb);
            wcur = wnext;
        }
    }
#pragma unroll
    for (int d = 0; d < R; ++d) {
        const int in = n_pad - R + min(d + 1, R - 1);
        const f32x4 wnext = lds[(in < w.n ? w.wbase + in * 64 : 0) + lane];
        mfma_chunk(acc0, acc1, wcur, ring[d]);
        wcur = wnext;
    }
    return acc0 + acc1;
}

#define MMQG_DSTAMP(slot)                                                                              \
    if (TRACE && tid == 0) a.trace[((size_t)blockIdx.x * a.T + t) * 8 + (slot)] = wall_clock64();

template <bool TRACE>
__global__ __launch_bounds__(kThreads, 2) void decoder_persist_fwd_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B, H = a.H, Cw = a.Cw;
    const int NU = H / 4;                          // units of 4 hidden units = 16 gate columns per layer
    const int g = blockIdx.x;
    const bool wg_l0 = g < NU, wg_l12 = g >= NU && g < 2 * NU;
    const int unit = wg_l0 ? g : g - NU;
    // score tiles (16 score columns x K = H): s_split workgroups per tile, each with 4 / s_split blocks of 16 questions
    // (rbw per workgroup) and its waves k-split 8 / rbw ways — as many of the layer-0 workgroups as fit take part
    const int n_stile = (a.S + 15) / 16;
    const int s_split = 4 * n_stile <= NU ? 4 : (2 * n_stile <= NU ? 2 : 1);
    const bool wg_s = g < n_stile * s_split;
    const int s_tile = g / s_split, s_rbw = 4 / s_split, s_kparts = kWaves / s_rbw;
    const int s_rb = (g % s_split) * s_rbw + wave % s_rbw, s_kp = wave / s_rbw;
    const int slot_f = kRows * H;                  // floats per (layer, token) slot of hx / xd
    // Every (layer, token) has its OWN slot in the exchange buffers: a line is written once (write-through) and only read
    // after the barrier behind its production, so no cache on the chip can hold an older copy of it — the loads are
    // plain cached loads, and the 16 workgroups of an XCD that all need the same h fetch it once from the fabric and 15
    // times from their L2 (with two alternating slots the loads had to bypass the L2s: 16 MB over the fabric per phase).
    auto hoff = [=](int l, int t) { return (l * (T + 1) + t + 1) * slot_f * 4; };           // h_l(t); t = -1: the initial state
    auto xdoff = [=](int l, int t) { return a.xd_off + (l * T + t) * slot_f * 4; };         // dropped copy of h_l(t), l < 2
    auto cxoff = [=](int t) { return a.cx_off + t * kRows * Cw * 4; };
    auto sxoff = [=](int t) { return a.sx_off + t * kRows * a.ldS * 4; };
    const int nch0 = (Cw + H) / 16, nch12 = 2 * H / 16, nchs = H / 16;

    // ---- LDS: [0,64) zero chunk | weight fragments | k-split partial tiles [8][64] | attention weights [7][kMaxSeg] floats
    int wb0 = 64, wb1 = 64, wbs = 64, wtotal = 64;
    if (wg_l0) { wb0 = wtotal; wtotal += nch0 * 64; }
    if (wg_l12) { wb0 = wtotal; wtotal += nch12 * 64; wb1 = wtotal; wtotal += nch12 * 64; }
    if (wg_s) { wbs = wtotal; wtotal += nchs * 64; }
    f32x4* scratch = lds + wtotal;                                   // [kWaves][64]
    float* att_e = reinterpret_cast<float*>(scratch + kWaves * 64);  // [7][kMaxSeg]
    f32x4* att_comb = reinterpret_cast<f32x4*>(att_e + 7 * kMaxSeg); // [6][16] partial contexts of the later waves of split items
    if (tid < 64) lds[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.hx, 0, a.ex_bytes, 0x00020000);

    // ---- layer-phase roles: wave = (row block mb = wave & 3, k half ks = wave >> 2); the ks == 1 wave of a row block
    // (waves 4-7: thread 0, which polls the barrier, is in none of them) keeps the recurrent half of its product, formed
    // ahead in an idle window, in registers and does the cell update: lane (j = lane & 15: row, q = lane >> 4: hidden unit 4 * unit + q) holds the four gates
    const int j = lane & 15, q = lane >> 4;
    const int mb = wave & 3, ks = wave >> 2;
    const int row = mb * 16 + j;
    const bool cellw = ks == 1;
    const int lane_off = (q * kRows + row) * 16;
    // state of (layer A, row, unit q) and, on the L1/L2 workgroups, of (layer 2, row, unit q)
    float hA = 0.f, cA = 0.f, hB = 0.f, cB = 0.f;
    float biasA[4] = {0.f, 0.f, 0.f, 0.f}, biasB[4] = {0.f, 0.f, 0.f, 0.f};
    int len = T;
    const int lA = wg_l0 ? 0 : 1;
    if ((wg_l0 || wg_l12) && cellw) {
        const int u = 4 * unit + q;
        const int64_t hstr = a.h0_stride_l ? a.h0_stride_l : (int64_t)B * H;
        if (row < B) {
            hA = a.h0[lA * hstr + (int64_t)row * H + u]; cA = a.c0[lA * hstr + (int64_t)row * H + u];
            if (wg_l12) { hB = a.h0[2 * hstr + (int64_t)row * H + u]; cB = a.c0[2 * hstr + (int64_t)row * H + u]; }
            if (a.lens) len = a.lens[row];
        }
        if (wg_l12) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                biasA[r] = a.b_ih1[r * H + u] + a.b_hh1[r * H + u];
                biasB[r] = a.b_ih2[r * H + u] + a.b_hh2[r * H + u];
            }
        }
        // h(-1) of the layer(s) into slot 1 of the exchange buffer
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(row < B ? hA : 0.f), rs, hoff(lA, -1) + ((unit * kRows + row) * 4 + q) * 4, 0, 16);
        if (wg_l12)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(row < B ? hB : 0.f), rs, hoff(2, -1) + ((unit * kRows + row) * 4 + q) * 4, 0, 16);
    }

    // h(-1) is on its way: once every wave's stores are acknowledged the launch announces itself at the flat start-up barrier,
    // which turns while the weights are read — when it has turned, every workgroup's h(-1) is published, so no second
    // barrier is needed before the first token
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gb::Ctx bar;
    gb::init_arrive(bar, a.bar, a.max_spins);

    // ---- weights -> LDS in fragment order: chunk c, lane (i = lane & 15: output column, kq = lane >> 4): 4 consecutive k
    // of that output column's weight row.  Gate column i of a unit: weight row (i & 3) * H + 4 * unit + (i >> 2).
    if (wg_l0) {
        for (int idx = tid; idx < nch0 * 64; idx += kThreads) {
            const int c = idx >> 6, l = idx & 63, i = l & 15, kq = l >> 4;
            const int row = (i & 3) * H + 4 * unit + (i >> 2), k = 16 * c + 4 * kq;
            const float* src = k < Cw ? a.w_ih0c + (int64_t)row * a.ld_w0 + k : a.w_hh0 + (int64_t)row * H + (k - Cw);
            lds[wb0 + idx] = *reinterpret_cast<const f32x4*>(src);
        }
    }
    if (wg_l12) {
        for (int idx = tid; idx < 2 * nch12 * 64; idx += kThreads) {
            const int which = idx >= nch12 * 64, id2 = idx - which * nch12 * 64;
            const int c = id2 >> 6, l = id2 & 63, i = l & 15, kq = l >> 4;
            const int row = (i & 3) * H + 4 * unit + (i >> 2), k = 16 * c + 4 * kq;
            const float* wi = which ? a.w_ih2 : a.w_ih1;
            const float* wh = which ? a.w_hh2 : a.w_hh1;
            const float* src = k < H ? wi + (int64_t)row * H + k : wh + (int64_t)row * H + (k - H);
            lds[(which ? wb1 : wb0) + id2] = *reinterpret_cast<const f32x4*>(src);
        }
    }
    if (wg_s) {
        for (int idx = tid; idx < nchs * 64; idx += kThreads) {
            const int c = idx >> 6, l = idx & 63, i = l & 15, kq = l >> 4;
            const int srow = min(16 * s_tile + i, a.S - 1);
            lds[wbs + idx] = *reinterpret_cast<const f32x4*>(a.w_attn_h + (int64_t)srow * a.ld_wa + 16 * c + 4 * kq);
        }
    }

    // ---- attention-phase roles: waves 1-4 = two text items (two waves each: rows halved; ONE item over all four waves when the batch
    // has no more than one per workgroup), waves 5-6 = two video items (or one, rows halved),
    // wave 7 = one audio item (32 columns); wave 0, whose lane 0 polls the barriers, has none (its first poll would wait
    // for the value rows it had in flight).  Item lists are question-major and a workgroup's two items of a kind lie G apart, so that the items of any batch
    // spread evenly over the CUs;
    // the host only takes shapes whose items fit one round (2 text + 2 video + 1 audio item per workgroup).
    const int chunks_t = (a.v.H + 63) / 64, chunks_v = (a.v.Dv + 63) / 64, chunks_a = (a.v.Da + 31) / 32;
    int it_mod = -1, it_half = 0, it_nhalf = 1, it_b = 0, it_chunk = 0;      // (it_half of it_nhalf: this wave's share of the item's rows)
    {
        int item = 0;
        if (wave >= 1 && wave <= 4) {
            // small batches (one text item per workgroup at most): all four text waves share it, a quarter of the rows each
            if (B * chunks_t <= a.G) { item = g; it_half = wave - 1; it_nhalf = 4; }
            else { item = g + ((wave - 1) >> 1) * a.G; it_half = (wave - 1) & 1; it_nhalf = 2; }
            if (item < B * chunks_t) it_mod = 0;
        }
        else if (wave == 5 || wave == 6) {
            if (B * chunks_v <= a.G) { item = g; it_half = wave - 5; it_nhalf = 2; }
            else { item = g + (wave - 5) * a.G; }
            if (item < B * chunks_v) it_mod = 2;
        }
        else if (wave == 7) { item = g; if (item < B * chunks_a) it_mod = 1; }
        const int nchunks = it_mod == 0 ? chunks_t : (it_mod == 2 ? chunks_v : chunks_a);
        it_b = item / nchunks; it_chunk = item - it_b * nchunks;
    }

    // question blocks (of 16) whose scores this workgroup's attention items read: bit r = block r (lane 0 polls for all waves)
    unsigned need_rb = 0;
    {
        const int nt = B * chunks_t, nv = B * chunks_v, na = B * chunks_a;
        if (B * chunks_t <= a.G) { if (g < nt) need_rb |= 1u << ((g / chunks_t) >> 4); }
        else { for (int k = 0; k < 2; ++k) if (g + k * a.G < nt) need_rb |= 1u << (((g + k * a.G) / chunks_t) >> 4); }
        if (B * chunks_v <= a.G) { if (g < nv) need_rb |= 1u << ((g / chunks_v) >> 4); }
        else { for (int k = 0; k < 2; ++k) if (g + k * a.G < nv) need_rb |= 1u << (((g + k * a.G) / chunks_v) >> 4); }
        if (g < na) need_rb |= 1u << ((g / chunks_a) >> 4);
    }
    bool ok = gb::init_wait(bar, a.expect_wg);        // (its workgroup barrier also closes the weight fill)
    f32x4 aheadA = f32x4{0.f, 0.f, 0.f, 0.f}, aheadB = aheadA;
    if (ok && wg_l0 && cellw)          // token 0 of layer 0: h_0(-1) W_hh0^T
        aheadA = wave_product<kRingAhead>(rs, make_prod(H / 16, 0, 1, hoff(0, -1), wb0 + (Cw / 16) * 64, lane_off), lds, lane);
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const uint64_t seed = a.drop ? eff_seed(a.seed, a.seed_off) : 0;

    for (int t = 0; ok && t < T; ++t) {
        MMQG_DSTAMP(0)
        // =========================================================== S: scores(t) into sx
        if (wg_s) {
            const int b = s_rb * 16 + j, col = 16 * s_tile + 4 * q;
            const bool st = s_kp == 0 && b < B && col < a.ldS;
            f32x4 pre = f32x4{0.f, 0.f, 0.f, 0.f};
            if (st) pre = *reinterpret_cast<const f32x4*>(a.pre_scores + ((int64_t)t * B + b) * a.ldS + col);
            const WaveProd w = make_prod(nchs, s_kp, s_kparts, hoff(2, t - 1), wbs, (q * kRows + b) * 16);      // h_top(t-1)
            f32x4 acc = s_kparts >= 4 ? wave_product<8>(rs, w, lds, lane) : wave_product<kRingLate>(rs, w, lds, lane);
            if (s_kp != 0) scratch[wave * 64 + lane] = acc;
            __syncthreads();
            if (st) {
                for (int kp = 1; kp < s_kparts; ++kp) acc += scratch[(kp * s_rbw + wave) * 64 + lane];
                stx(rs, sxoff(t) + (b * a.ldS + col) * 4, acc + pre);
            }
        }
        MMQG_DSTAMP(1)
        // The attention items of question b need the score rows of b's block of 16 questions only — the tiles of at most
        // n_stile workgroups — not the whole chip: the score workgroups count their arrival per question block
        // (point-to-point: stores drained by every wave, then ONE lane adds), and a workgroup's lane 0 polls the counters
        // of the blocks its items belong to.  (a.scnt null: the device-wide barrier, as before.)
        if (a.scnt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wg_s && tid == 0) {
                const int rb0 = (g % s_split) * s_rbw;
                for (int r = 0; r < s_rbw; ++r) gb::add_rlx(a.scnt + (rb0 + r) * gb::kLine, 1u);
            }
        } else {
            gb::arrive(bar);
        }
        // idle window of every wave with an attention item: the first 16 KB of its value rows are on their way while the score
        // tiles finish and the barrier turns (they do not depend on the scores)
        constexpr int kU = 8;
        f32x4 cur[kU], nxt[kU];
        int L = 0, D = 0, seg_off = 0, ctx_off = 0, valid = 0, r_lo = 0, r_hi = 0, col = 0, lanes = 16, groups = 4, rgp = 0, last = 0, coff = 0, rowb = 0;
        bool col_ok = false, stream = false;
        const float* base = a.v.text;
        if (it_mod >= 0) {
            const int b = it_b;
            const int cwidth = it_mod == 1 ? 32 : 64;
            if (it_mod == 0) { base = a.v.text + (int64_t)b * a.v.text_stride_b; L = a.v.Lt; D = a.v.H; seg_off = 0; ctx_off = 0; valid = a.v.text_len ? a.v.text_len[b] : L; }
            else if (it_mod == 1) { base = a.v.audio + (int64_t)b * a.v.audio_stride_b; L = a.v.Lav; D = a.v.Da; seg_off = a.v.Lt; ctx_off = a.v.H; valid = a.v.av_len ? a.v.av_len[b] : L; }
            else { base = a.v.video + (int64_t)b * a.v.video_stride_b; L = a.v.Lav; D = a.v.Dv; seg_off = a.v.Lt + a.v.Lav; ctx_off = a.v.H + a.v.Da; valid = a.v.av_len ? a.v.av_len[b] : L; }
            const int n_stream = a.v.zero_past_len ? max(1, min(L, valid)) : L;
            const int rows_half = (n_stream + it_nhalf - 1) / it_nhalf;       // rows of this wave: all, or its half
            r_lo = it_half * rows_half; r_hi = min(n_stream, r_lo + rows_half);
            lanes = cwidth / 4;                                               // float4 column lanes (16, or 8 for audio)
            groups = 64 / lanes;                                              // row groups (4, or 8)
            rgp = lane / lanes;
            col = it_chunk * cwidth + 4 * (lane % lanes);
            col_ok = col < D;
            last = max(r_hi - 1, r_lo);
            stream = col_ok && r_hi > r_lo;
            coff = col * 4; rowb = D * 4;
        }
        // (the question's value rows through a buffer descriptor: one 32-bit offset per load instead of a 64-bit address)
        const auto rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, L * D * 4, 0x00020000);
        auto fetch = [&rv, coff, rowb, last, groups](f32x4 (&dst)[kU], int first) {
#pragma unroll
            for (int u = 0; u < kU; ++u)
                dst[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, coff + min(first + u * groups, last) * rowb, 0, 0));
        };
        if (stream) { fetch(cur, r_lo + rgp); fetch(nxt, r_lo + rgp + kU * groups); }
        if (a.scnt) {
            bool okp = true;
            if (tid == 0) {
                const unsigned target = (unsigned)n_stile * (unsigned)(t + 1);
                for (int r = 0; r < 4 && okp; ++r)
                    if ((need_rb >> r) & 1) okp = gb::spin_until_ge(a.scnt + r * gb::kLine, target, a.bar->fail, bar.max_spins);
            }
            ok = __syncthreads_and(okp);
        } else {
            ok = gb::wait(bar);
        }
        if (!ok) break;
        MMQG_DSTAMP(2)
        // =========================================================== ATT: softmax + contexts of step t
        {
            f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
            float inv = 0.f;
            float* e = att_e + max(wave - 1, 0) * kMaxSeg;
            if (it_mod >= 0) {
                const bool masked = a.v.mask_mode == MMQG_MASK_INTENDED;
                // softmax of the whole segment (both halves of a split item do it)
                // (the segment's scores — up to kMaxSeg / 64 = 5 per lane — all asked for before the first is used: as a loop of
                // one load per trip this was five L2 round trips in a row right behind the score barrier, 3 us of the
                // attention window)
                float lmax = -INFINITY;
                {
                    constexpr int NI = kMaxSeg / 64;
                    float sv[NI];
                    const int sbase = sxoff(t) + (it_b * a.ldS + seg_off) * 4;
#pragma unroll
                    for (int u = 0; u < NI; ++u)
                        sv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, sbase + min(lane + 64 * u, L - 1) * 4, 0, 0));
#pragma unroll
                    for (int u = 0; u < NI; ++u) {
                        const int i = lane + 64 * u;
                        if (i < L) {
                            const float s = (masked && i >= valid) ? -INFINITY : sv[u];
                            e[i] = s;
                            lmax = fmaxf(lmax, s);
                        }
                    }
                }
                lmax = wave_max(lmax);
                float lsum = 0.f;
                for (int i = lane; i < L; i += 64) {
                    const float x = expf(e[i] - lmax);
                    e[i] = x;
                    lsum += x;
                }
                lsum = wave_sum(lsum);
                inv = 1.0f / lsum;
                __builtin_amdgcn_wave_barrier();
                if (it_chunk == 0 && it_half == 0) {
                    float* arow = a.attn + ((int64_t)t * B + it_b) * a.ldS + seg_off;
                    for (int i = lane; i < L; i += 64) arow[i] = e[i] * inv;
                }
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                if (stream) {
                    for (int r = r_lo + rgp; r < r_hi; r += kU * groups) {
                        float wv[kU];
#pragma unroll
                        for (int u = 0; u < kU; ++u) { const int i = r + u * groups; wv[u] = i < r_hi ? e[i] : 0.f; }
#pragma unroll
                        for (int u = 0; u < kU; u += 2) { acc0 += wv[u] * cur[u]; acc1 += wv[u + 1] * cur[u + 1]; }
#pragma unroll
                        for (int u = 0; u < kU; ++u) cur[u] = nxt[u];
                        if (r + 2 * kU * groups < r_hi) fetch(nxt, r + 2 * kU * groups);
                    }
                }
                acc0 += acc1;
                // combine the row groups of the wave (lanes with equal column lane)
                for (int off = lanes; off < 64; off <<= 1) {
                    acc0.x += __shfl_xor(acc0.x, off, 64); acc0.y += __shfl_xor(acc0.y, off, 64);
                    acc0.z += __shfl_xor(acc0.z, off, 64); acc0.w += __shfl_xor(acc0.w, off, 64);
                }
                part = acc0;
                if (it_half > 0 && lane < 16) att_comb[(wave - 1) * 16 + lane] = part;       // (waves 2..6)
            }
            __syncthreads();                       // (every wave, with or without an item)
            if (it_mod >= 0 && it_half == 0 && lane < lanes && col_ok) {
                for (int k = 1; k < it_nhalf; ++k) part += att_comb[(wave - 1 + k) * 16 + lane];
                part = part * inv;
                *reinterpret_cast<f32x4*>(a.ctx + ((int64_t)t * B + it_b) * Cw + ctx_off + col) = part;      // saved for backward
                stx(rs, cxoff(t) + (((ctx_off + col) >> 2) * kRows + it_b) * 16, part);                        // operand of layer 0
            }
        }
        MMQG_DSTAMP(3)
        gb::arrive(bar);
        // idle window of the layer-1/2 workgroups (they have nothing to do until layer 0 is through): the recurrent half of
        // layer 2's product, h_2(t-1) W_hh2^T
        if (wg_l12 && cellw) aheadB = wave_product<kRingAhead>(rs, make_prod(H / 16, 0, 1, hoff(2, t - 1), wb1 + (H / 16) * 64, lane_off), lds, lane);
        ok = gb::wait(bar);
        if (!ok) break;
        MMQG_DSTAMP(4)
        // =========================================================== L0 / L1 / L2
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const bool mine = ph == 0 ? wg_l0 : wg_l12;
            float st_g[4] = {0.f, 0.f, 0.f, 0.f}, st_hd = 0.f;
            bool st_on = false;
            if (mine) {
                // the late half of the product (layer 0: the contexts; layers 1, 2: the layer below), k-split over the two
                // waves of a row block
                const int l = ph;
                const int off_x = l == 0 ? cxoff(t) : (a.drop ? xdoff(l - 1, t) : hoff(l - 1, t));
                const int wbase = l == 2 ? wb1 : wb0;
                f32x4 pre4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (l == 0 && cellw && row < B) {
                    const float* g0 = a.gates + ((int64_t)t * B + row) * 4 * H + 4 * unit + q;
                    pre4 = f32x4{g0[0], g0[H], g0[2 * H], g0[3 * H]};
                }
                f32x4 acc;
                if (l == 0) acc = wave_product<12>(rs, make_prod(Cw / 16, ks, 2, off_x, wbase, lane_off), lds, lane);
                else acc = wave_product<kRingLate>(rs, make_prod(H / 16, ks, 2, off_x, wbase, lane_off), lds, lane);
                if (!cellw) scratch[wave * 64 + lane] = acc;
                __syncthreads();
                if (cellw) {
                    const f32x4 sum = acc + scratch[(wave - 4) * 64 + lane] + pre4 + (l == 2 ? aheadB : aheadA);
                    const int u = 4 * unit + q;
                    const bool valid = row < B, active = valid && t < len;
                    float& hreg = l == 2 ? hB : hA;
                    float& creg = l == 2 ? cB : cA;
                    const float* bias = l == 2 ? biasB : biasA;
                    const float gi = sigmoidf_(sum.x + bias[0]), gf = sigmoidf_(sum.y + bias[1]);
                    const float gg = tanhf(sum.z + bias[2]), go = sigmoidf_(sum.w + bias[3]);
                    if (active) { creg = gf * creg + gi * gg; hreg = go * tanhf(creg); }
                    float hd = 0.f;
                    const bool to_above = l < 2;
                    if (a.drop && to_above && active)
                        hd = hreg * dropout_scale(seed, a.stream_base + (uint64_t)l * T + t, (uint64_t)((int64_t)row * H + u), a.drop_p);
                    const int uoff = ((unit * kRows + row) * 4) * 4;
                    {
                        const float hv = valid ? hreg : 0.f;
                        const f32x4 pk = f32x4{hv, __shfl(hv, j + 16, 64), __shfl(hv, j + 32, 64), __shfl(hv, j + 48, 64)};
                        if (q == 0) stx(rs, hoff(l, t) + uoff, pk);
                    }
                    if (a.drop && to_above) {
                        const f32x4 pk = f32x4{hd, __shfl(hd, j + 16, 64), __shfl(hd, j + 32, 64), __shfl(hd, j + 48, 64)};
                        if (q == 0) stx(rs, xdoff(l, t) + uoff, pk);
                    }
                    st_g[0] = active ? gi : 0.f; st_g[1] = active ? gf : 0.f; st_g[2] = active ? gg : 0.f; st_g[3] = active ? go : 0.f;
                    st_hd = hd; st_on = valid;
                }
            }
            MMQG_DSTAMP(5 + ph)
            gb::arrive(bar);
            // saved activations (only later kernels read them): behind the arrival
            if (mine && cellw && st_on) {
                const int l = ph, u = 4 * unit + q;
                float* grow = a.gates + (((int64_t)l * T + t) * B + row) * 4 * H + u;
                grow[0] = st_g[0]; grow[H] = st_g[1]; grow[2 * H] = st_g[2]; grow[3 * H] = st_g[3];
                const int64_t e = (((int64_t)l * (T + 1) + t + 1) * B + row) * H + u;
                a.hs[e] = l == 2 ? hB : hA; a.cs[e] = l == 2 ? cB : cA;
                if (a.drop && l < 2) a.hdrop[(((int64_t)l * T + t) * B + row) * H + u] = st_hd;
            }
            // idle windows: the recurrent half of the NEXT product of this workgroup's layer, from an h that an earlier
            // barrier published — layer 1's h_1(t-1) W_hh1^T while layer 0 runs, layer 0's h_0(t) W_hh0^T (for token
            // t + 1) while layer 1 runs
            if (ph == 0 && wg_l12 && cellw)
                aheadA = wave_product<kRingAhead>(rs, make_prod(H / 16, 0, 1, hoff(1, t - 1), wb0 + (H / 16) * 64, lane_off), lds, lane);
            if (ph == 1 && wg_l0 && cellw && t + 1 < T)
                aheadA = wave_product<kRingAhead>(rs, make_prod(H / 16, 0, 1, hoff(0, t), wb0 + (Cw / 16) * 64, lane_off), lds, lane);
            ok = gb::wait(bar);
            if (!ok) break;
        }
    }
    if (!ok) {
        gb::report_failure(a.sticky_fail, a.host_fail);
        if (tid == 0) a.poison[0] = __builtin_nanf("");
    }
}
#undef MMQG_DSTAMP

inline int64_t align_up(int64_t v, int64_t al) { return (v + al - 1) / al * al; }

struct WsLayout { int64_t bar, scnt, hx, xd, cx, sx, sticky, total; };
WsLayout ws_layout(int T, int H, int Cw, int ldS) {
    WsLayout w;
    w.bar = 0;
    w.scnt = align_up((int64_t)sizeof(gb::XBar), 128);                  // 4 counters, 128 bytes apart (zero-filled with the barrier block)
    w.hx = align_up(w.scnt + 4 * 128, 256);
    w.xd = w.hx + (int64_t)3 * (T + 1) * kRows * H * 4;
    w.cx = w.xd + (int64_t)2 * T * kRows * H * 4;
    w.sx = w.cx + (int64_t)T * kRows * Cw * 4;
    w.sticky = align_up(w.sx + (int64_t)T * kRows * ldS * 4, 256);
    w.total = w.sticky + 256;
    return w;
}

int lds_need(int H, int Cw) {
    const int l0 = ((Cw + H) / 16 + H / 16) * 1024;             // layer-0 unit + a score tile
    const int l12 = 2 * (2 * H / 16) * 1024;
    return 1024 + std::max(l0, l12) + kWaves * 1024 + 7 * kMaxSeg * 4 + 6 * 16 * 16;
}

}  // namespace

namespace mmqg {

static int g_dec_persist_launches = 0;
int decoder_persist_launch_count() { return g_dec_persist_launches; }
static unsigned long long* g_dtrace_buf = nullptr;
static int64_t g_dtrace_words = 0;
void decoder_persist_set_trace(unsigned long long* buf, int64_t words) { g_dtrace_buf = buf; g_dtrace_words = buf ? words : 0; }

bool decoder_persist_shape_ok(const mmqg_decoder_seq& d) {
    static const bool off = [] {
        const char* e = getenv("MMQG_NO_PERSIST");
        const char* f = getenv("MMQG_NO_PERSIST_DEC");
        return (e && atoi(e) != 0) || (f && atoi(f) != 0);
    }();
    if (off) return false;
    const mmqg_attn_values& v = d.values;
    const int H = d.H, Cw = v.H + v.Da + v.Dv, S = v.Lt + 2 * v.Lav;
    if (d.L != 3 || d.T < 2 || d.B < 1 || d.B > kRows) return false;
    if (H < 128 || H % 16 || Cw % 16 || v.H != H || v.H % 4 || v.Da % 4 || v.Dv % 4 || d.E % 4 || d.ld_attn % 4) return false;
    if (v.Lt > kMaxSeg || v.Lav > kMaxSeg) return false;
    if ((S + 15) / 16 > H / 4) return false;          // the score tiles live on the layer-0 workgroups
    {   // the attention items of a token must fit one round: 2 text + 2 video + 1 audio item per workgroup
        const int G = std::min(persist_device_cus(), 256);
        // (a CU-masked stream with fewer CUs fails the same test again at launch time: a.G is what the items are laid out over)
        if (d.B * ((v.H + 63) / 64) > 2 * G || d.B * ((v.Dv + 63) / 64) > 2 * G || d.B * ((v.Da + 31) / 32) > G) return false;
    }
    return lds_need(H, Cw) <= kLdsBudget;
}

int64_t decoder_persist_ws_bytes(const mmqg_decoder_seq& d) {
    if (!decoder_persist_shape_ok(d)) return 0;
    persist_runtime_prepare();
    const mmqg_attn_values& v = d.values;
    return ws_layout(d.T, d.H, v.H + v.Da + v.Dv, d.ld_attn).total;
}

// 0 = done (the whole time loop), 1 = not taken, < 0 = error
int decoder_seq_fwd_persistent(const mmqg_decoder_seq& d, hipStream_t s) {
    if (!d.persist_ws || !decoder_persist_shape_ok(d)) return 1;
    const mmqg_attn_values& v = d.values;
    const int T = d.T, B = d.B, H = d.H, E = d.E, Cw = v.H + v.Da + v.Dv, S = v.Lt + 2 * v.Lav;
    const WsLayout wl = ws_layout(T, H, Cw, d.ld_attn);
    if (wl.sticky - wl.hx >= (int64_t)1 << 31) return 1;
    if (d.persist_ws_bytes < wl.total || !aligned16(d.persist_ws)) return 1;
    const int G = std::min(persist_usable_cus(s, false), 256);
    if (G < 2 * (H / 4)) return 1;
    if (B * ((v.H + 63) / 64) > 2 * G || B * ((v.Dv + 63) / 64) > 2 * G || B * ((v.Da + 31) / 32) > G) return 1;
    const bool drop = d.training && d.dropout_p > 0.f;
    if (drop && !d.hdrop) return 1;
    const float* ptrs[] = {d.w_attn, d.w_ih[0], d.w_hh[0], d.w_ih[1], d.w_hh[1], d.w_ih[2], d.w_hh[2], d.scores, d.ctx, d.gates,
                           v.text, v.audio, v.video};
    for (const float* p : ptrs) if (!p || !aligned16(p)) return 1;
    if ((E + H) % 4 || (E + Cw) % 4 || v.text_stride_b % 4 || v.audio_stride_b % 4 || v.video_stride_b % 4) return 1;
    if ((v.mask_mode == MMQG_MASK_INTENDED || v.zero_past_len) && !(v.text_len && v.av_len)) return 1;
    const int lds_bytes = lds_need(H, Cw);
    static int attr_set = 0;
    if (attr_set == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_persist_fwd_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_persist_fwd_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e != hipSuccess) (void)hipGetLastError();
        attr_set = e == hipSuccess ? 1 : -1;
    }
    if (attr_set < 0) return 1;
    {
        static int occ_lds = -1, occ_ok = 0;
        if (occ_lds != lds_bytes) {
            int nb = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &nb, reinterpret_cast<const void*>(decoder_persist_fwd_kernel<false>), kThreads, (size_t)lds_bytes);
            if (e != hipSuccess) (void)hipGetLastError();
            occ_lds = lds_bytes; occ_ok = (e == hipSuccess && nb >= 1) ? 1 : 0;
        }
        if (!occ_ok) return 1;
    }
    if (persist_begin(s) != 0) return 1;

    char* ws = reinterpret_cast<char*>(d.persist_ws);
    const int64_t BH = (int64_t)B * H;
    // (slot 0 of hs / cs already holds the initial state: decoder_seq_fwd) the barrier block starts from zero.  The
    // exchange buffers need no initial value: every slot is written before it is read, and rows >= B, which nobody writes,
    // only feed MFMA output rows >= B, which nobody stores.
    const CopySeg init{reinterpret_cast<float*>(ws), nullptr, wl.hx / 4};
    MMQG_TRY(copy_or_zero_multi(&init, 1, s));
    DecArgs a{};
    a.T = T; a.B = B; a.H = H; a.E = E; a.Cw = Cw; a.S = S; a.ldS = d.ld_attn; a.G = G;
    a.v = v;
    a.pre_scores = d.scores;
    a.w_attn_h = d.w_attn + E; a.ld_wa = E + H;
    a.w_ih0c = d.w_ih[0] + E; a.ld_w0 = E + Cw; a.w_hh0 = d.w_hh[0];
    a.w_ih1 = d.w_ih[1]; a.w_hh1 = d.w_hh[1]; a.w_ih2 = d.w_ih[2]; a.w_hh2 = d.w_hh[2];
    a.b_ih1 = d.b_ih[1]; a.b_hh1 = d.b_hh[1]; a.b_ih2 = d.b_ih[2]; a.b_hh2 = d.b_hh[2];
    a.h0 = d.h0; a.c0 = d.c0; a.h0_stride_l = d.h0_stride_l; a.lens = d.lens;
    a.gates = d.gates; a.hs = d.hs; a.cs = d.cs; a.hdrop = d.hdrop; a.attn = d.attn; a.ctx = d.ctx;
    a.drop = drop ? 1 : 0; a.drop_p = drop ? d.dropout_p : 0.f; a.seed = d.seed; a.stream_base = d.stream_base; a.seed_off = d.seed_offset;
    a.hx = reinterpret_cast<float*>(ws + wl.hx);
    a.xd_off = (int)(wl.xd - wl.hx); a.cx_off = (int)(wl.cx - wl.hx); a.sx_off = (int)(wl.sx - wl.hx);
    a.ex_bytes = (int)(wl.sticky - wl.hx);
    a.bar = reinterpret_cast<gb::XBar*>(ws + wl.bar);
    // MMQG_DEC_S_P2P=1 (opt-in A/B): per-question-block arrival counters behind the score phase instead of the
    // device-wide barrier.  Measured at config 2 on one box (round 4): the wait behind the score tiles 5.38 us either way,
    // attention window 14.14 (barrier) / 14.30 us (counters), token 41.3 us both: what a workgroup waits for there is the
    // score workgroups' own chain (h_top load round + tile + store acknowledgement), not the barrier's 2.5 us.
    static const bool s_p2p = [] { const char* e = getenv("MMQG_DEC_S_P2P"); return e && atoi(e) != 0; }();
    a.scnt = s_p2p ? reinterpret_cast<unsigned*>(ws + wl.scnt) : nullptr;
    a.poison = d.hs + ((int64_t)2 * (T + 1) + T) * BH;
    a.sticky_fail = reinterpret_cast<unsigned*>(ws + wl.sticky);
    a.host_fail = persist_host_fail_word();
    a.expect_wg = (unsigned)(G + persist_test_extra_wg());
    a.max_spins = persist_test_max_spins() ? persist_test_max_spins() : gb::kDefaultSpins;
    a.trace = nullptr;
    if (g_dtrace_buf && (int64_t)G * T * 8 <= g_dtrace_words) a.trace = g_dtrace_buf;
    if (a.trace) hipLaunchKernelGGL(decoder_persist_fwd_kernel<true>, dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a);
    else hipLaunchKernelGGL(decoder_persist_fwd_kernel<false>, dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a);
    g_dec_persist_launches += 1;
    persist_end(s);
    return check_launch("decoder_persist_fwd");
}

}  // namespace mmqg
