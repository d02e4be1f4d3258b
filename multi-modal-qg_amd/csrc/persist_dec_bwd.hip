// Persistent BACKWARD time loop of the attention decoder: autograd of AttnDecoder.forward (model/decoder.py:74-107) as
// train.py:177's loss.backward() runs it through the T teacher-forced steps of train.py:171-175 — ONE launch instead of
// five dependent launches per token (sequence.hip, decoder_seq_bwd).
//
// A token is a strict chain (the attention feeds h_top(t-1) back into token t, so no layer wavefront exists):
//     dh_2(t) = dhtop(t) + dG_2(t+1) W_hh2 + dS(t+1) W_attn_h            -> cell_2 backward -> dG_2(t)
//     dh_1(t) = dG_1(t+1) W_hh1 + mask_1(t) * (dG_2(t) W_ih2)             -> cell_1 backward -> dG_1(t)
//     dh_0(t) = dG_0(t+1) W_hh0 + mask_0(t) * (dG_1(t) W_ih1)             -> cell_0 backward -> dG_0(t)
//     dctx(t) = dG_0(t) W_ih0[:, E:]                                       -> attention backward -> dS(t)
// Every product has K = 4H (2,048) and few output columns, so it is cut along K: slice s of a layer holds 32 hidden
// units x 4 gates = 128 k.  Workgroup (s, c) of the H/32 x G/(H/32) grid keeps the [128 k] x [its column tiles] blocks
// of the three layers' matrices in LDS for the whole sequence (config 2: 15 tiles x 8 KB = 120 KB per CU, 30 MB in all,
// read from memory once).  What keeps ONE device-wide barrier per stage although the k-slices must meet: the CONSUMER
// sums them.  A stage of layer l is, per workgroup:
//     sum the NS partial tiles of dh_l(t) for the slice's 32 units (128 KB of 16-byte loads, one wave-load = 1 KB)
//     -> cell backward for 64 rows x 32 units (redundantly on the G/NS workgroups of the slice: it is cheap, and dc and
//        the pass-through part of dh live in registers for the whole sequence)
//     -> dG fragment [64 rows x 128 k] into LDS in MFMA operand order
//     -> late product  dG_l(t) W_ih_l   (the operand the NEXT stage waits for)  -> partial tiles, write-through
//     -> arrive at the barrier
//     -> ahead product dG_l(t) W_hh_l   (needed a whole token later) in the window while the barrier turns
// so the chain carries half of the arithmetic and one barrier per stage: five barriers per token
//     SW  dS(t+1) W_attn_h  (K = 488: k-ranges x column tiles, partial tiles)
//     P2, P1, P0            (as above; P0's late product is dctx, stored row-major per slice)
//     ATT dctx(t) = sum of P0's slices; attention backward d(attn)_i = V_i . dctx, dS_i = a_i (d(attn)_i - ctx . dctx)
//         (one wave per (question, modality, row range), four rows per 16-lane group at a time, the first two batches of
//         value rows in flight while the barrier turns) — the 54 MB stream of the token.
// Exchange data has ONE SLOT PER TOKEN: written once with write-through (sc1) stores, read only behind the barrier
// after its production, so plain cached loads cannot see an older copy (as in persist_dec.hip); the barrier is the
// fence-free XCD-hierarchical one of grid_barrier.h.  Weight / bias gradients, the value gradients and dxemb stay in
// the hoisted grouped GEMMs of decoder_seq_bwd: this kernel leaves dgates, dscores, dctx and the initial-state
// gradients exactly where the launched loop leaves them.
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
constexpr int kU = 32;                   // hidden units per k-slice (x 4 gates = 128 k = 8 chunks of 16)
constexpr int kChunks = 4 * kU / 16;     // 8
constexpr int kMaxNS = 16;               // slices (H <= 512); the kernel is instantiated for NS = 4, 8, 16 (H = 128, 256, 512)
constexpr int kMaxKR = 8;                // k-ranges of the score-gradient product
constexpr int kSkip = -1;                // load_early: the ahead product is not asked for yet
constexpr int kDirQ = 16;                // direct score-gradient product: at most 16 quads (64 positions) of a dS row per cell lane
constexpr int kMaxD = 512;               // widest value row (16 lanes x 8 x 16 bytes)
constexpr int kLdsBudget = 160 * 1024 - 256;      // (the grid barrier's 256 static bytes)
constexpr int kFragF4 = kChunks * 4 * kRows;      // float4 of the dG fragment: 32 KB
constexpr int kTraceSlots = 16;

struct DbArgs {
    int T, B, H, Cw, S, ldS, ldD, G;
    int NS, GS, RB;                          // k-slices per layer, workgroups per slice, 16-row blocks that hold rows (1, 2, 4)
    int tpcL2, tpcL1, tpcL0, tpcA;           // late / ahead column tiles per workgroup
    int nchS, NKR, cpk;                      // score-gradient product: k-chunks, k-ranges, chunks per range
    int cpq, n_t, n_v, n_a;                  // attention plan: workgroups per question, wave-items per modality
    int direct;                              // the score-gradient product runs inside the top layer's cell stage (no SW stage)
    int dbg;                                 // diagnostics (MMQG_PDB_DBG): 1 = wave 0 never loads ahead of its poll, 2 = no dgates stores (WRONG results), 4 = print the plan once
    mmqg_attn_values v;
    const float* w_ihT1; const float* w_ihT2;                        // [H][4H] k-major copies
    const float* w_hhT0; const float* w_hhT1; const float* w_hhT2;   // [H][4H]
    const float* w_ih0cT;                                            // [Cw][4H]
    const float* w_attn_hT;                                          // [H][ldS]
    const float* gates; const float* cs;                             // saved [3][T][B][4H], [3][T+1][B][H]
    const float* attn; const float* ctx;                             // saved [T][B][ldS], [T][B][Cw]
    const float* dhtop;                                              // [T][B][H]
    const int32_t* lens;
    float* dgates; float* dscores; float* dctx;                      // out [3][T][B][4H], [T][B][ldD], [T][B][Cw]
    float* dh_out; float* dc_out;                                    // out [3][B][H]
    float drop_p; int drop; uint64_t seed, stream_base; const int32_t* seed_off;
    // exchange (one buffer descriptor): per token  P2 | P1 | P0x | D0 | SW | R2 | R1 | R0 (the reduced ahead products)
    float* ex; int ex_bytes;
    int tok_stride, p1_off, p0x_off, d0_off, sw_off, r_off;          // bytes
    gb::XBar* bar;
    float* poison; unsigned* sticky_fail; unsigned* host_fail; unsigned expect_wg, max_spins;
    unsigned long long* trace;
};

__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const int32_t* off) {
    return off ? seed + (uint64_t)(uint32_t)off[0] * 0x9E3779B97F4A7C15ull : seed;
}

// dropout keep-scales of the four elements idx .. idx+3 (idx a multiple of 4): the one Philox block dropout_scale()
// would compute for each of them
__device__ __forceinline__ f32x4 dropout_scale4(uint64_t seed, uint64_t stream_id, uint64_t idx, float p) {
    uint32_t r[4];
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(idx >> 2), (uint32_t)(idx >> 34), (uint32_t)stream_id,
                  (uint32_t)(stream_id >> 32), r);
    const float keep = 1.0f / (1.0f - p);
    f32x4 o;
    o.x = (float)(r[0] >> 8) * (1.0f / 16777216.0f) < p ? 0.f : keep;
    o.y = (float)(r[1] >> 8) * (1.0f / 16777216.0f) < p ? 0.f : keep;
    o.z = (float)(r[2] >> 8) * (1.0f / 16777216.0f) < p ? 0.f : keep;
    o.w = (float)(r[3] >> 8) * (1.0f / 16777216.0f) < p ? 0.f : keep;
    return o;
}

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 tanh4(const f32x4& v) { return f32x4{tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)}; }
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// sum over the 16 lanes that share lane >> 4 (a DPP row), in every lane: quad_perm [1,0,3,2], [2,3,0,1], then row_ror 4, 8
// — no LDS crossbar, no index registers (ds_bpermute needed four of them per call site in a kernel that has none to spare)
__device__ __forceinline__ float sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    return v;
}

template <typename Rsrc>
__device__ __forceinline__ f32x4 ldx(const Rsrc& rs, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
// the same with the wave-uniform part of the offset in an SGPR (no address VGPR per load).  The range check of a raw
// buffer covers the VGPR offset only: voff = kOob makes the load return zero without a branch.
constexpr int kOob = 0x7fffff00;
template <typename Rsrc>
__device__ __forceinline__ f32x4 ldxs(const Rsrc& rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane(soff), 0));
}
template <typename Rsrc>
__device__ __forceinline__ void stx(const Rsrc& rs, int off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 16);      // write-through (sc1)
}

__device__ __forceinline__ void mfma_chunk(f32x4& acc0, f32x4& acc1, const f32x4& wt, const f32x4& x) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.x, x.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.y, x.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.z, x.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.w, x.w, acc1, 0, 0, 0);
}

// what a cell lane has in flight before the chain operand arrives: saved activations of (layer, t, row, 4 units),
// dhtop (top layer) and the (already reduced) ahead product of token t + 1
struct Early {
    f32x4 gi, gf, gg, go, cprev, cnew, extra, ahead;
};

template <typename Rsrc>
__device__ __forceinline__ void load_early(Early& e, const Rsrc& rs, const float* gr, const float* cr, const float* xr, int H,
                                           int64_t BH, int aoff) {
    e.gi = *reinterpret_cast<const f32x4*>(gr); e.gf = *reinterpret_cast<const f32x4*>(gr + H);
    e.gg = *reinterpret_cast<const f32x4*>(gr + 2 * H); e.go = *reinterpret_cast<const f32x4*>(gr + 3 * H);
    e.cprev = *reinterpret_cast<const f32x4*>(cr); e.cnew = *reinterpret_cast<const f32x4*>(cr + BH);
    e.extra = xr ? *reinterpret_cast<const f32x4*>(xr) : zero4();
    if (aoff != kSkip) e.ahead = ldx(rs, aoff);      // (first token: an out-of-range offset, reads as zero)
}

// Sum of the NS slices' partial tiles of an ahead product (dG_l(t) W_hh_l, [H/4 column quads][64 rows][4], contiguous per
// slice) into the reduced block the cells of token t - 1 read: this workgroup's n_out 16-byte outputs starting at o0, by ONE
// wave in a window behind a barrier arrival (lane = (output, half of the slices); 8 loads in flight per lane at NS = 16).
// In two halves, so that the wave's share of the window's product runs while the loads are in flight.
template <int NS>
struct AheadSum {
    f32x4 v[NS / 2];
    int o, n_out, dst;
    template <typename Rsrc>
    __device__ __forceinline__ void issue(const Rsrc& rs, int src0, int sstride, int dst0, int o0, int ob, int n, int lane) {
        constexpr int HALF = NS / 2;
        const int half = lane >> 5;
        o = ob + (lane & 31); n_out = n; dst = dst0 + (o0 + o) * 16;
        const int voff = o < n_out ? (o0 + o) * 16 + half * HALF * sstride : kOob;     // (the lane's half of the slices: per-lane part)
#pragma unroll
        for (int i = 0; i < HALF; ++i) v[i] = ldxs(rs, voff, src0 + i * sstride);
    }
    template <typename Rsrc>
    __device__ __forceinline__ void finish(const Rsrc& rs, int lane) {
        f32x4 sum = v[0];
#pragma unroll
        for (int i = 1; i < NS / 2; ++i) sum += v[i];
        sum.x += __shfl_xor(sum.x, 32, 64); sum.y += __shfl_xor(sum.y, 32, 64);
        sum.z += __shfl_xor(sum.z, 32, 64); sum.w += __shfl_xor(sum.w, 32, 64);
        if ((lane >> 5) == 0 && o < n_out) stx(rs, dst, sum);
    }
};
template <int NS, typename Rsrc>
__device__ __forceinline__ void reduce_ahead(const Rsrc& rs, int src0, int sstride, int dst0, int o0, int n_out, int lane, int ob0 = 0) {
    for (int ob = ob0; ob < n_out; ob += 32) {
        AheadSum<NS> r;
        r.issue(rs, src0, sstride, dst0, o0, ob, n_out, lane);
        r.finish(rs, lane);
    }
}

// One wave's share of the products of the dG fragment (LDS, MFMA operand order) with `ntile` resident weight tiles:
// tiles part, part + nparts, ... ; tile tl's 16 x 16 result (lane (j, q): row rb * 16 + j, columns 4 q .. 4 q + 3) goes to
// byte offset off0 + tl * tile_step of the exchange buffer, write-through.
template <typename Rsrc>
__device__ __forceinline__ void products(const Rsrc& rs, const f32x4* frag, const f32x4* wl, int ntile, int part, int nparts,
                                         int rb, int j, int q, int off0, int tile_step) {
    if (part < 0 || part >= ntile) return;
    f32x4 x[kChunks];
#pragma unroll
    for (int c = 0; c < kChunks; ++c) x[c] = frag[(c * 4 + q) * kRows + rb * 16 + j];
    for (int tl = part; tl < ntile; tl += nparts) {
        f32x4 acc0 = zero4(), acc1 = zero4();
        const f32x4* w = wl + tl * 512;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) mfma_chunk(acc0, acc1, w[c * 64], x[c]);
        stx(rs, off0 + tl * tile_step, acc0 + acc1);
    }
}

// ---- attention backward of one wave-item: rows [r0, r1) of a question's segment, D floats wide.  A wave's four 16-lane
// groups take 8 >> LOG rows each per batch, NKP = 1 << LOG 16-byte columns per lane and row (16 lanes x NKP x 16 bytes >= D):
// every one of the 8 loads a lane has in flight per batch carries data for any width — with one row per group a 128-float
// audio row used 2 of the 8 loads and its wave ran three times longer than a text wave of the same byte count.
constexpr int kAttK = kMaxD / 64;        // 8 loads per lane and batch
template <int LOG, typename Rsrc>
__device__ __forceinline__ void att_fetch(f32x4 (&dst)[kAttK], const Rsrc& rv, int first, int rg, int cl, int last, int D, int Dq, int oob) {
    constexpr int NKP = 1 << LOG, RPL = kAttK >> LOG;
#pragma unroll
    for (int k = 0; k < kAttK; ++k) {
        const int kk = k >> LOG, cq = cl + 16 * (k & (NKP - 1));
        const int r = min(first + rg * RPL + kk, last);
        dst[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, cq < Dq ? (r * D + 4 * cq) * 4 : oob, 0, 0));
    }
}
template <int LOG>
__device__ __forceinline__ float att_fetch_a(const float* arow, int first, int rg, int cl, int r1) {
    constexpr int RPL = kAttK >> LOG;
    const int r = first + rg * RPL + cl;
    return (cl < RPL && r < r1) ? arow[r] : 0.f;
}
// one batch: d(attn) of 4 * RPL rows from the value rows in `buf`, dS stored.  The dctx columns come from LDS at every use
// (8 ds_read_b128 per batch at most): kept in registers they cost 32 VGPRs in a kernel that has none to spare.
template <int LOG>
__device__ __forceinline__ void att_batch(const f32x4 (&buf)[kAttK], const f32x4* vec, int Dq, float aw, float* dsr, float dot, int r,
                                          int r1, int n_stream, int rg, int cl) {
    constexpr int NKP = 1 << LOG, RPL = kAttK >> LOG;
    float mine = 0.f;
#pragma unroll
    for (int kk = 0; kk < RPL; ++kk) {
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < NKP; ++k) p += dot4(buf[kk * NKP + k], vec[min(cl + 16 * k, Dq - 1)]);      // (columns past D: the value loads returned zero)
        p = sum16(p);
        mine = cl == kk ? p : mine;
    }
    const int ri = r + rg * RPL + cl;
    if (cl < RPL && ri < r1) {
        const float da = ri < n_stream ? mine : 0.f;
        __hip_atomic_store(dsr + ri, aw * (da - dot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through
    }
}
// the row loop of an item: two batches in flight, the buffers take turns (no register copies)
template <int LOG, typename Rsrc>
__device__ __forceinline__ void att_stream(f32x4 (&b0)[kAttK], f32x4 (&b1)[kAttK], float a0, float a1, const Rsrc& rv,
                                           const float* arow, float* dsr, const f32x4* vec, float dot, int r0, int r1, int n_stream,
                                           int rg, int cl, int last, int D, int Dq, int oob) {
    constexpr int RPL = kAttK >> LOG, STEP = 4 * RPL;
    for (int r = r0; r < r1; r += 2 * STEP) {
        att_batch<LOG>(b0, vec, Dq, a0, dsr, dot, r, r1, n_stream, rg, cl);
        if (r + 2 * STEP < r1) { att_fetch<LOG>(b0, rv, r + 2 * STEP, rg, cl, last, D, Dq, oob); a0 = att_fetch_a<LOG>(arow, r + 2 * STEP, rg, cl, r1); }
        if (r + STEP < r1) {
            att_batch<LOG>(b1, vec, Dq, a1, dsr, dot, r + STEP, r1, n_stream, rg, cl);
            if (r + 3 * STEP < r1) { att_fetch<LOG>(b1, rv, r + 3 * STEP, rg, cl, last, D, Dq, oob); a1 = att_fetch_a<LOG>(arow, r + 3 * STEP, rg, cl, r1); }
        }
    }
}

// stamps go to LDS (behind the dG fragment; 64 bytes per token) and to memory only when the loop is over: a global store
// per stamp sat in wave 0's memory queue in front of its loads and changed what it measured
#define MMQG_GSTAMP(slot)                                                                              \
    if (TRACE && tid == 0) stamps[t * kTraceSlots + (slot)] = (unsigned)wall_clock64();

template <bool TRACE, int NS, bool DIRECT>
__global__ __launch_bounds__(kThreads, 2) void decoder_persist_bwd_kernel(DbArgs a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B, H = a.H, Cw = a.Cw, GS = a.GS, NKR = a.NKR;
    const int tok_stride = a.tok_stride, p1_off = a.p1_off, p0x_off = a.p0x_off, d0_off = a.d0_off, sw_off = a.sw_off, r_off = a.r_off;
    const int tpcL2 = a.tpcL2, tpcL1 = a.tpcL1, tpcL0 = a.tpcL0, tpcA = a.tpcA;
    const int g = blockIdx.x;
    const int sl = g % NS, cg = g / NS;                 // this workgroup's k-slice and its index among the slice's workgroups
    const int r_bytes = H * 256;                        // one reduced block: [H/4][64 rows][16 bytes]
    const int red_n = (16 * H + (int)gridDim.x - 1) / (int)gridDim.x, red_o0 = min(16 * H, g * red_n);      // this workgroup's outputs of a reduction
    const int red_cnt = min(red_n, 16 * H - red_o0);
    const bool cellwg = cg < GS;
    const int dg_rows = (kRows + GS - 1) / GS;          // rows of the gate gradients each of a slice's workgroups stores
    const int nlate21 = H / 16, nlate0 = Cw / 16, nahead = H / 16;
    const int pq = 2 * H / 4;                           // column quads of a P2 / P1 slice block (late | ahead)

    // ---- LDS: weight blocks of P2 | P1 | P0 (late tiles then ahead tiles each) | SW chunks | dG fragment (aliased: the
    // attention stage's dctx vectors, the score-gradient stage's k-part combine)
    const int wb2 = 0, wb1 = wb2 + (tpcL2 + tpcA) * 512, wb0 = wb1 + (tpcL1 + tpcA) * 512, wbs = wb0 + (tpcL0 + tpcA) * 512;
    // direct mode: [wave][k] quads W_attn_h[k][this wave's 4 units] in place of the SW stage's chunks
    constexpr bool direct = DIRECT;
    const int wbd = wbs;
    const int Sd = (a.S + 3) & ~3;            // positions a cell lane multiplies (k >= S: zero weights, never past ldS <= ldD)
    f32x4* frag = lds + wbs + (direct ? kWaves * Sd : a.cpk * 64);
    unsigned* stamps = reinterpret_cast<unsigned*>(frag + kFragF4);      // [T][kTraceSlots], stamped instantiation only
    if (TRACE) for (int i = tid; i < a.T * kTraceSlots; i += kThreads) stamps[i] = 0u;
    const unsigned t_entry = TRACE ? (unsigned)wall_clock64() : 0u;      // (slot 15 of the first token: kernel entry, before the weights are read)
    gb::Ctx bar;
    gb::init_arrive(bar, a.bar, a.max_spins);       // the census / flat barrier turns while the 120 KB of weights are on their way

    // my tiles of a layer: late tiles [cg * tpcL, ...) of nlate, ahead tiles [cg * tpcA, ...) of H/16
    const int nL2 = cellwg ? max(0, min(tpcL2, nlate21 - cg * tpcL2)) : 0;
    const int nL1 = cellwg ? max(0, min(tpcL1, nlate21 - cg * tpcL1)) : 0;
    const int nL0 = cellwg ? max(0, min(tpcL0, nlate0 - cg * tpcL0)) : 0;
    const int nA = cellwg ? max(0, min(tpcA, nahead - cg * tpcA)) : 0;

    // ---- weights -> LDS in fragment order: tile, chunk c, lane (i = lane & 15: output column of the tile, kq = lane >> 4):
    // the four k = 16 c + 4 kq + {0..3} of the slice, i.e. gate c >> 1, units 32 sl + 16 (c & 1) + 4 kq + {0..3}: one
    // 16-byte load from the k-major copy WT[n][gate * H + unit].  The items of all six blocks form ONE list that is read
    // sixteen 16-byte loads per thread at a time, in SOURCE order — 8 consecutive lanes cover the 128 contiguous bytes of a
    // (column, gate) pair, i.e. whole cache lines — and scattered into the LDS (120 KB per workgroup, 30 MB per launch: a
    // loop of one load + one store per thread and matrix took ~60 us of the launch).
    {
        const float* srcs[6] = {a.w_ihT2, a.w_hhT2, a.w_ihT1, a.w_hhT1, a.w_ih0cT, a.w_hhT0};
        const int wbase[6] = {wb2, wb2 + tpcL2 * 512, wb1, wb1 + tpcL1 * 512, wb0, wb0 + tpcL0 * 512};
        const int first[6] = {cg * tpcL2, cg * tpcA, cg * tpcL1, cg * tpcA, cg * tpcL0, cg * tpcA};
        const int count[6] = {nL2, nA, nL1, nA, nL0, nA};
        int start[7];
        start[0] = 0;
#pragma unroll
        for (int m = 0; m < 6; ++m) start[m + 1] = start[m] + count[m] * 512;
        constexpr int kUn = 16;       // (15 items per thread at config 2: one round of loads)
        for (int base = 0; base < start[6]; base += kUn * kThreads) {
            f32x4 v[kUn];
            int dst[kUn];
#pragma unroll
            for (int u = 0; u < kUn; ++u) {
                const int it = base + u * kThreads + tid;
                int m = 0;
#pragma unroll
                for (int mm = 1; mm < 6; ++mm) m = it >= start[mm] ? mm : m;
                const float* src = m == 0 ? srcs[0] : (m == 1 ? srcs[1] : (m == 2 ? srcs[2] : (m == 3 ? srcs[3] : (m == 4 ? srcs[4] : srcs[5]))));
                const int st = m == 0 ? start[0] : (m == 1 ? start[1] : (m == 2 ? start[2] : (m == 3 ? start[3] : (m == 4 ? start[4] : start[5]))));
                const int fi = m == 0 ? first[0] : (m == 1 ? first[1] : (m == 2 ? first[2] : (m == 3 ? first[3] : (m == 4 ? first[4] : first[5]))));
                const int wbm = m == 0 ? wbase[0] : (m == 1 ? wbase[1] : (m == 2 ? wbase[2] : (m == 3 ? wbase[3] : (m == 4 ? wbase[4] : wbase[5]))));
                const int loc = it - st, tl = loc >> 9, jj = loc & 511;
                const int u4 = jj & 7, gate = (jj >> 3) & 3, ni = jj >> 5;
                const bool on = it < start[6];
                dst[u] = on ? wbm + tl * 512 + (2 * gate + (u4 >> 2)) * 64 + (u4 & 3) * 16 + ni : -1;
                v[u] = on ? *reinterpret_cast<const f32x4*>(src + (int64_t)(16 * (fi + tl) + ni) * 4 * H + gate * H + kU * sl + 4 * u4) : zero4();
            }
#pragma unroll
            for (int u = 0; u < kUn; ++u)
                if (dst[u] >= 0) lds[dst[u]] = v[u];
        }
    }
    // score-gradient product: workgroup (sl, cg) owns column tile 2 sl + (cg & 1) and k-range cg >> 1
    const int sw_tile = 2 * sl + (cg & 1), sw_kr = cg >> 1;
    const int sw_c0 = sw_kr * a.cpk;
    const int sw_n = (!direct && cellwg && sw_kr < NKR) ? max(0, min(a.cpk, a.nchS - sw_c0)) : 0;
    if (direct) {
        // (rows k >= S of the transposed copy are zero by contract)
        for (int idx = tid; idx < kWaves * Sd; idx += kThreads) {
            const int w = idx / Sd, k = idx - w * Sd;
            const float* src = a.w_attn_hT + (int64_t)(kU * sl + 4 * w) * a.ldS + k;
            lds[wbd + idx] = f32x4{src[0], src[a.ldS], src[2 * a.ldS], src[3 * a.ldS]};
        }
    }
    for (int idx = tid; idx < sw_n * 64; idx += kThreads) {
        const int c = idx >> 6, l = idx & 63, i = l & 15, kq = l >> 4;
        const int k = 16 * (sw_c0 + c) + 4 * kq, n = 16 * sw_tile + i;
        f32x4 w = zero4();
        if (k < a.ldS) w = *reinterpret_cast<const f32x4*>(a.w_attn_hT + (int64_t)n * a.ldS + k);
        // (rows k >= S of the transposed copy are zero by contract; columns past ldS do not exist)
        lds[wbs + idx] = w;
    }

    const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.ex, 0, a.ex_bytes, 0x00020000);
    // (direct mode reads the dS rows themselves; quads past the row through an out-of-range offset)
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(a.dscores, 0, T * B * a.ldD * 4, 0x00020000);
    const int nq_dir = Sd / 4;

    // ---- cell-lane role: wave = unit quad (units 32 sl + 4 wave .. + 3), lane = row
    const int u0 = kU * sl + 4 * wave;
    int len = T;
    if (a.lens && lane < B) len = a.lens[lane];
    f32x4 dc2 = zero4(), dhc2 = zero4(), dc1 = zero4(), dhc1 = zero4(), dc0 = zero4(), dhc0 = zero4();

    // ---- product role: row block rb, column part cp of KP
    const int RB = a.RB, KP = kWaves / RB;
    const int rb = wave % RB, cp = wave / RB;
    const bool rb_on = rb * 16 < B;

    // ---- attention plan (closed form): question qb = g % B on the cpq workgroups g = qb + B * qq; item k = 7 qq + wave - 1:
    // text parts [0, n_t), video parts [n_t, n_t + n_v), audio parts [n_t + n_v, n_t + n_v + n_a)
    const int qb = g % B, qq = g / B;
    const bool attwg = qq < a.cpq;
    const int k_lo = 7 * qq, k_hi = k_lo + 7;
    int it_mod = -1, it_r0 = 0, it_r1 = 0;          // this wave's item: modality (0 text, 1 audio, 2 video), rows [r0, r1)
    if (attwg && wave >= 1) {
        const int k = k_lo + wave - 1;
        int part = -1, nparts = 1, L = 0;
        if (k < a.n_t) { it_mod = 0; part = k; nparts = a.n_t; L = a.v.Lt; }
        else if (k < a.n_t + a.n_v) { it_mod = 2; part = k - a.n_t; nparts = a.n_v; L = a.v.Lav; }
        else if (k < a.n_t + a.n_v + a.n_a) { it_mod = 1; part = k - a.n_t - a.n_v; nparts = a.n_a; L = a.v.Lav; }
        if (it_mod >= 0) {
            const int per = (L + nparts - 1) / nparts;
            it_r0 = min(L, part * per); it_r1 = min(L, it_r0 + per);
            if (it_r1 <= it_r0) it_mod = -1;
        }
    }
    // which dctx vectors this workgroup sums (one per modality it holds an item of) and which it writes to global memory:
    // threads 128 m .. 128 m + 127 take modality m
    const int vm = wave >> 1;
    bool v_need = false, v_write = false;
    int v_D = 0, v_off = 0;
    if (attwg && vm < 3) {
        const int lo = vm == 0 ? 0 : (vm == 2 ? a.n_t : a.n_t + a.n_v);
        const int hi = vm == 0 ? a.n_t : (vm == 2 ? a.n_t + a.n_v : a.n_t + a.n_v + a.n_a);
        v_need = hi > lo && lo < k_hi && hi > k_lo;
        v_write = v_need && lo >= k_lo && lo < k_hi;
        v_D = vm == 0 ? a.v.H : (vm == 1 ? a.v.Da : a.v.Dv);
        v_off = vm == 0 ? 0 : (vm == 1 ? a.v.H : a.v.H + a.v.Da);
    }

    // (announced before the weights were read; no second barrier: the first stage reads nothing another workgroup of this
    // launch has written — saved activations only — and ends with an arrival of its own)
    bool ok = gb::init_wait(bar, a.expect_wg);
    const uint64_t seed = a.drop ? eff_seed(a.seed, a.seed_off) : 0;
    const int64_t BH = (int64_t)B * H;
    bool pend = false;                               // an arrival whose wait is still to come

// Per-lane indices are re-derived from an opaque copy of the lane id in every stage: left to itself hipcc hoists every
// per-lane address of the five stages out of the token loop, runs out of registers and then spills each load result
// the moment it arrives (one s_waitcnt vmcnt(0) per load).
#define MMQG_LANE_ROLES                                                               \
    int ln = lane;                                                                    \
    asm volatile("; lane roles" : "+v"(ln));                                          \
    const int row = ln, rowc = min(ln, B - 1), j = ln & 15, q = ln >> 4;              \
    const bool rvalid = ln < B;                                                       \
    (void)row; (void)rowc; (void)j; (void)q; (void)rvalid;

    if (TRACE && tid == 0) stamps[(T - 1) * kTraceSlots + 15] = t_entry;
    for (int t = T - 1; ok && t >= 0; --t) {
        MMQG_GSTAMP(0)
        const int tok = t * tok_stride, tok1 = (t + 1) * tok_stride;
        // =================================================== SW: partial tiles of dS(t+1) W_attn_h (feeds dh_2(t))
        if (!direct && t + 1 < T) {
            if (pend) { ok = gb::wait(bar); pend = false; }
            if (!ok) break;
            MMQG_GSTAMP(1)
            MMQG_LANE_ROLES
            if (sw_n > 0 && rb_on) {
                // wave (rb, cp): chunks cp, cp + KP, ... of this workgroup's k-range; lane (j: row, q: k quad)
                // (the wave's dS quads all asked for before the first product: one L2 round trip, not one per chunk)
                const int dsrow = ((t + 1) * B + min(rb * 16 + j, B - 1)) * a.ldD * 4;
                constexpr int kPre = 4;
                f32x4 xs[kPre];
#pragma unroll
                for (int u = 0; u < kPre; ++u) {
                    const int c = cp + u * KP, k = 16 * (sw_c0 + c) + 4 * q;
                    xs[u] = ldx(rd, (c < sw_n && k < a.ldS) ? dsrow + 4 * k : kOob);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
                for (int u = 0; u < kPre; ++u) {
                    const int c = cp + u * KP;
                    if (c < sw_n) mfma_chunk(acc0, acc1, lds[wbs + c * 64 + ln], xs[u]);
                }
                for (int c = cp + kPre * KP; c < sw_n; c += KP) {
                    const int k = 16 * (sw_c0 + c) + 4 * q;
                    const f32x4 x = ldx(rd, k < a.ldS ? dsrow + 4 * k : kOob);
                    mfma_chunk(acc0, acc1, lds[wbs + c * 64 + ln], x);
                }
                frag[wave * 64 + ln] = acc0 + acc1;
            }
            __syncthreads();
            if (sw_n > 0 && cp == 0 && rb_on) {
                f32x4 sum = frag[wave * 64 + ln];
                for (int p = 1; p < KP; ++p) sum += frag[(rb + RB * p) * 64 + ln];
                stx(rs, tok1 + sw_off + ((sw_tile * NKR + sw_kr) * 4 + q) * 1024 + (rb * 16 + j) * 16, sum);
            }
            MMQG_GSTAMP(2)
            gb::arrive(bar);
            pend = true;
            // window: layer 0's ahead product of token t + 1 (stored in P0's window, drained before the attention stage's
            // barrier), summed over the slices for cell (0, t)
            if (wave == kWaves - 1 && red_cnt > 0)
                reduce_ahead<NS>(rs, tok1 + p0x_off, (H / 4) * 1024, tok1 + r_off, red_o0, red_cnt, ln);
        }
        // =================================================== P2 / P1 / P0
#pragma unroll
        for (int ph_i = 0; ph_i < 3; ++ph_i) {
            const int l = 2 - ph_i;
            f32x4& dcl = l == 2 ? dc2 : (l == 1 ? dc1 : dc0);
            f32x4& dhcl = l == 2 ? dhc2 : (l == 1 ? dhc1 : dhc0);
            const int pl_off = l == 2 ? 0 : (l == 1 ? p1_off : p0x_off);            // this layer's own exchange block
            const int above_off = l == 1 ? 0 : p1_off;                              // the layer above's (late operand), l < 2
            const int a_quads = l == 0 ? H / 4 : pq;                                // column quads per slice of the block with the ahead tiles
            const int a_first = l == 0 ? 0 : nlate21 * 4;                           // first ahead quad inside it
            MMQG_LANE_ROLES
            Early e;
            f32x4 dG0 = zero4(), dG1 = zero4(), dG2 = zero4(), dG3 = zero4();
            // ---- early: saved activations (+ dhtop for the top layer) and the reduced ahead product of token t + 1; wave 0
            // (its lane 0 polls the barrier) asks for them behind the wait
            const float* gr = a.gates + (((int64_t)l * T + t) * B + rowc) * 4 * H + u0;
            const float* cr = a.cs + (((int64_t)l * (T + 1) + t) * B + rowc) * H + u0;
            const float* xr = l == 2 ? a.dhtop + ((int64_t)t * B + rowc) * H + u0 : nullptr;
            const int aoff = t + 1 < T ? tok1 + r_off + l * r_bytes + ((8 * sl + wave) * kRows + row) * 16 : kOob;
            // (direct mode: one barrier fewer per token lies between the window that sums the ahead product and this stage —
            // the sum is only complete at the barrier this stage waits for, so it is asked for with the late operand)
            const int aoff_early = direct ? kSkip : aoff;
            // (... unless this workgroup arrived last in its XCC: that lane 0 then releases the XCC, and its first poll would
            // wait for the loads)
            const bool early_now = cellwg && (wave != 0 || !pend || (!(a.dbg & 1) && !__builtin_amdgcn_readfirstlane((int)bar.leader)));
            if (early_now) load_early(e, rs, gr, cr, xr, H, BH, aoff_early);
            if (pend) { ok = gb::wait(bar); pend = false; }
            MMQG_GSTAMP(3 + 3 * ph_i)
            if (cellwg && !early_now) load_early(e, rs, gr, cr, xr, H, BH, aoff_early);
            if (!ok) break;
            if (cellwg) {
                // ---- the late operand: partial tiles of the product that came down the chain
                constexpr int NLATE = kDirQ > NS ? kDirQ : NS;
                f32x4 late[NLATE];
                if (direct) e.ahead = ldx(rs, aoff);
                if (l == 2 && direct) {
                    // this row of dS(t+1), 16 bytes at a time (the first token has none: zeros)
                    const int base = ((t + 1) * B + rowc) * a.ldD * 4;
                    const int n = t + 1 < T ? nq_dir : 0;
#pragma unroll
                    for (int i = 0; i < kDirQ; ++i) late[i] = ldx(rd, i < n ? base + 16 * i : kOob);
                } else if (l == 2) {
                    // (k-ranges past NKR, and the first token, which has no dS(t+1): out-of-range offsets read as zero)
                    const int base = tok1 + sw_off + (((2 * sl + (wave >> 2)) * NKR) * 4 + (wave & 3)) * 1024;
                    const int n = t + 1 < T ? NKR : 0;
#pragma unroll
                    for (int i = 0; i < kMaxKR; ++i) late[i] = ldxs(rs, i < n ? row * 16 : kOob, base + i * 4096);
                } else {
                    const int base = tok + above_off + (8 * sl + wave) * 1024;
#pragma unroll
                    for (int i = 0; i < NS; ++i) late[i] = ldxs(rs, row * 16, base + i * pq * 1024);
                }
                // (all of them in flight before the first is used: left to itself the scheduler pairs every load with its add —
                // a load, s_waitcnt vmcnt(0), an add, sixteen times over, i.e. sixteen L2 round trips in a row on the chain)
                __builtin_amdgcn_sched_barrier(0);
                f32x4 pi = zero4();
                const f32x4 ph = e.ahead;
                if (l == 2 && direct) {
                    // dS(t+1)[row][:] W_attn_h[:][this wave's 4 units]: the weights are wave-uniform LDS reads
                    const f32x4* dw = lds + wbd + wave * Sd;
#pragma unroll
                    for (int i = 0; i < kDirQ; ++i)
                        if (i < nq_dir) {
                            pi += late[i].x * dw[4 * i] + late[i].y * dw[4 * i + 1];
                            pi += late[i].z * dw[4 * i + 2] + late[i].w * dw[4 * i + 3];
                            asm volatile("" ::: "memory");      // (four weight quads at a time: hoisted, the 64 reads take 256 registers)
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < (l == 2 ? kMaxKR : NS); ++i) pi += late[i];
                }
                if (l < 2 && a.drop)
                    pi *= dropout_scale4(seed, a.stream_base + (uint64_t)l * T + t, (uint64_t)((int64_t)row * H + u0), a.drop_p);
                // ---- cell backward of (row, 4 units)
                if (rvalid) {
                    f32x4 dh = dhcl + ph + pi;
                    if (t < len) {
                        if (l == 2) dh += e.extra;
                        const f32x4 tc = tanh4(e.cnew);
                        const f32x4 dct = dcl + dh * e.go * (1.f - tc * tc);
                        dG0 = dct * e.gg * e.gi * (1.f - e.gi);
                        dG1 = dct * e.cprev * e.gf * (1.f - e.gf);
                        dG2 = dct * e.gi * (1.f - e.gg * e.gg);
                        dG3 = dh * tc * e.go * (1.f - e.go);
                        dcl = dct * e.gf;
                        dhcl = zero4();
                    } else {
                        dhcl = dh;               // finished row: its state was carried forward, so is its gradient
                    }
                }
                // fragment: k = gate * 32 + unit -> chunk 2 gate + (wave >> 2), k quad wave & 3: index (8 gate + wave) * 64 + row
                frag[(0 + wave) * kRows + row] = dG0; frag[(8 + wave) * kRows + row] = dG1;
                frag[(16 + wave) * kRows + row] = dG2; frag[(24 + wave) * kRows + row] = dG3;
            }
            __syncthreads();
            MMQG_GSTAMP(4 + 3 * ph_i)
            const int wb = l == 2 ? wb2 : (l == 1 ? wb1 : wb0);
            const int tpcL = l == 2 ? tpcL2 : (l == 1 ? tpcL1 : tpcL0);
            const int nL = l == 2 ? nL2 : (l == 1 ? nL1 : nL0);
            // ---- late product: partial tiles for the next stage, exchange layout [slice][column quad][64 rows][4] (P2, P1)
            // or, for dctx, row-major [slice][64 rows][Cw]
            if (rb_on) {
                if (l > 0)
                    products(rs, frag, lds + wb + ln, nL, cp, KP, rb, j, q,
                             tok + pl_off + (sl * pq + 4 * cg * tpcL + q) * 1024 + (rb * 16 + j) * 16, 4 * 1024);
                else
                    products(rs, frag, lds + wb + ln, nL, cp, KP, rb, j, q,
                             tok + d0_off + ((sl * kRows + rb * 16 + j) * Cw + 16 * cg * tpcL + 4 * q) * 4, 16 * 4);
            }
            MMQG_GSTAMP(5 + 3 * ph_i)
            gb::arrive(bar);
            pend = true;
            // ---- window while the barrier turns: the ahead product dG_l(t) W_hh_l (needed by cell (l, t-1)); in P0's window
            // also the sum over the slices of layer 2's ahead product of this token for cell (2, t-1) — its tiles were stored
            // in P2's window and drained at the arrival to P1's barrier, which this workgroup has passed — by the last wave,
            // whose loads fly while it multiplies; then the gate gradients for the hoisted weight-gradient GEMMs (every
            // workgroup of the slice a few rows; wave 1 also stores wave 0's quad)
            const bool red_win = l == 0 || (direct && l == 2);      // windows in which the last wave sums an ahead product
            const bool red_on = red_win && wave == kWaves - 1 && red_cnt > 0;
            {
                // the window's product waves of this row block: not wave 0 (it polls), and in P0's window not the last wave
                // (it sums); their column parts are renumbered
                int part = -1, nparts = 0;
                for (int p = 0; p < KP; ++p) {
                    const int w = rb + RB * p;
                    const bool act = w != 0 && !(red_win && w == kWaves - 1 && red_cnt > 0);
                    if (w == wave && act) part = nparts;
                    nparts += act ? 1 : 0;
                }
                if (rb_on && part >= 0)
                    products(rs, frag, lds + wb + tpcL * 512 + ln, nA, part, nparts, rb, j, q,
                             tok + pl_off + (sl * a_quads + a_first + 4 * cg * tpcA + q) * 1024 + (rb * 16 + j) * 16, 4 * 1024);
            }
            if (red_on && l == 0) reduce_ahead<NS>(rs, tok + nlate21 * 4 * 1024, pq * 1024, tok + r_off + 2 * r_bytes, red_o0, red_cnt, ln);
            // (direct mode, P2's window: layer 0's ahead product of token t + 1 — stored in P0's window, drained before the
            // attention stage's barrier, which this workgroup has passed; the SW stage's window did this)
            if (red_on && l == 2 && t + 1 < T)
                reduce_ahead<NS>(rs, tok1 + p0x_off, (H / 4) * 1024, tok1 + r_off, red_o0, red_cnt, ln);
            // the gate gradients for the hoisted weight-gradient GEMMs: every workgroup of the slice stores dg_rows rows, ONE
            // wave reads them back from the fragment so that 8 lanes cover a whole 128-byte line (32 units of a gate); the
            // scattered 16-byte stores of the cell lanes took 2-3 us to be acknowledged, in front of the next stage's loads
            if (cellwg && wave == kWaves - 2 && !(a.dbg & 2)) {
                for (int idx = ln; idx < dg_rows * 32; idx += 64) {
                    const int r = cg * dg_rows + (idx >> 5), gt = (idx >> 3) & 3, wq = idx & 7;
                    if (r < B)
                        *reinterpret_cast<f32x4*>(a.dgates + (((int64_t)l * T + t) * B + r) * 4 * H + gt * H + kU * sl + 4 * wq) =
                            frag[(8 * gt + wq) * kRows + r];
                }
            }
        }
        if (!ok) break;
        // =================================================== ATT: dctx(t) from P0's slices, attention backward -> dS(t)
        {
            MMQG_LANE_ROLES
            f32x4 cur[kAttK], nxt[kAttK];
            float a_cur = 0.f, a_nxt = 0.f;
            const int rg = ln >> 4, cl = ln & 15;
            const int tq = (wave & 1) * 64 + ln;       // index among the 128 threads of a modality's dctx sum
            int L = 1, D = 4, seg_off = 0, n_stream = 0, oob = 16, last = 0, Dq = 1, lg = 3;
            const float* base = a.v.text;
            if (it_mod >= 0) {
                int valid;
                if (it_mod == 0) { base = a.v.text + (int64_t)qb * a.v.text_stride_b; L = a.v.Lt; D = a.v.H; seg_off = 0; valid = a.v.text_len ? a.v.text_len[qb] : L; }
                else if (it_mod == 1) { base = a.v.audio + (int64_t)qb * a.v.audio_stride_b; L = a.v.Lav; D = a.v.Da; seg_off = a.v.Lt; valid = a.v.av_len ? a.v.av_len[qb] : L; }
                else { base = a.v.video + (int64_t)qb * a.v.video_stride_b; L = a.v.Lav; D = a.v.Dv; seg_off = a.v.Lt + a.v.Lav; valid = a.v.av_len ? a.v.av_len[qb] : L; }
                n_stream = a.v.zero_past_len ? min(L, valid) : L;
                oob = L * D * 4; last = max(n_stream - 1, 0); Dq = D / 4;
                lg = D > 128 ? 3 : 1;       // 16-byte columns per lane and row: 8 (one row per 16-lane group and batch) or 2 (four rows)
            }
            // (the question's value rows through a buffer descriptor: out-of-range offsets read as zero)
            const auto rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, L * D * 4, 0x00020000);
            const float* arow = a.attn + ((int64_t)t * B + qb) * a.ldS + seg_off;
            if (it_mod >= 0) {         // (waves 1..7: nothing here depends on the chain) the first two batches of value rows
                if (lg == 3) { att_fetch<3>(cur, rv, it_r0, rg, cl, last, D, Dq, oob); att_fetch<3>(nxt, rv, it_r0 + 4, rg, cl, last, D, Dq, oob);
                               a_cur = att_fetch_a<3>(arow, it_r0, rg, cl, it_r1); a_nxt = att_fetch_a<3>(arow, it_r0 + 4, rg, cl, it_r1); }
                else { att_fetch<1>(cur, rv, it_r0, rg, cl, last, D, Dq, oob); att_fetch<1>(nxt, rv, it_r0 + 16, rg, cl, last, D, Dq, oob);
                       a_cur = att_fetch_a<1>(arow, it_r0, rg, cl, it_r1); a_nxt = att_fetch_a<1>(arow, it_r0 + 16, rg, cl, it_r1); }
            }
            // (the forward's context of this thread's dctx columns: the softmax Jacobian's row dot is ctx . dctx)
            f32x4 cxv = zero4();
            const float* cxp = a.ctx + ((int64_t)t * B + qb) * Cw + v_off + 4 * tq;
            const bool cx_on = v_need && 4 * tq < v_D;
            const bool cx_now = wave != 0 || !pend || !__builtin_amdgcn_readfirstlane((int)bar.leader);      // (as the cells' early loads)
            if (cx_on && cx_now) cxv = *reinterpret_cast<const f32x4*>(cxp);
            if (pend) { ok = gb::wait(bar); pend = false; }
            if (!ok) break;
            MMQG_GSTAMP(12)
            if (cx_on && !cx_now) cxv = *reinterpret_cast<const f32x4*>(cxp);
            // The carries of the top layer sit out this stage in the dG fragment, which is free between this stage's wait
            // (the workgroup has read it for the last time) and the next product: with two batches of value rows and the
            // slices' dctx parts in flight the stage was 4 registers short, and ONE spilled quad gives the kernel a scratch
            // allocation that every launch has to set up.
            f32x4* park = frag + 512;
            park[tid] = dc2; park[kThreads + tid] = dhc2;
            // dctx(t)[qb][modality columns] = sum over the slices' row-major partial blocks; 128 threads per modality
            float* dotp = reinterpret_cast<float*>(frag + 3 * 128);       // [8] per-wave parts of ctx . dctx
            if (v_need) {
                const int cq = tq;
                const bool on = 4 * cq < v_D;
                const int off = (qb * Cw + v_off + 4 * cq) * 4;
                f32x4 p[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) p[i] = ldxs(rs, on ? off : kOob, tok + d0_off + i * kRows * Cw * 4);
                f32x4 sum = zero4();
#pragma unroll
                for (int i = 0; i < NS; ++i) sum += p[i];
                if (on) {
                    frag[vm * 128 + cq] = sum;
                    if (v_write) *reinterpret_cast<f32x4*>(a.dctx + ((int64_t)t * B + qb) * Cw + v_off + 4 * cq) = sum;
                }
                const float part = wave_sum(dot4(sum, cxv));
                if (ln == 0) dotp[wave] = part;
            }
            __syncthreads();
            MMQG_GSTAMP(13)
            if (it_mod >= 0) {
                const float dot = dotp[2 * it_mod] + dotp[2 * it_mod + 1];
                float* dsr = a.dscores + ((int64_t)t * B + qb) * a.ldD + seg_off;
                const f32x4* vec = frag + it_mod * 128;
                if (lg == 3) att_stream<3>(cur, nxt, a_cur, a_nxt, rv, arow, dsr, vec, dot, it_r0, it_r1, n_stream, rg, cl, last, D, Dq, oob);
                else att_stream<1>(cur, nxt, a_cur, a_nxt, rv, arow, dsr, vec, dot, it_r0, it_r1, n_stream, rg, cl, last, D, Dq, oob);
            }
            dc2 = park[tid]; dhc2 = park[kThreads + tid];
            MMQG_GSTAMP(14)
            gb::arrive(bar);
            pend = true;
            // window: layer 1's ahead product of this token (stored in P1's window, drained before P0's barrier), summed
            // over the slices for cell (1, t-1)
            if (wave == kWaves - 1 && red_cnt > 0)
                reduce_ahead<NS>(rs, tok + p1_off + nlate21 * 4 * 1024, pq * 1024, tok + r_off + r_bytes, red_o0, red_cnt, ln);
        }
    }
    // ---- gradient of the initial state: carry + dG_l(0) W_hh_l (the ahead tiles of token 0) + dS(0) W_attn_h for the top layer
    if (ok) {
        const int t = 0;        // (stamps of the epilogue are not recorded)
        if (pend) { ok = gb::wait(bar); pend = false; }
        MMQG_LANE_ROLES
        if (ok) {
            if (sw_n > 0 && rb_on) {
                const float* ds = a.dscores + (int64_t)min(rb * 16 + j, B - 1) * a.ldD;
                f32x4 acc0 = zero4(), acc1 = zero4();
                for (int c = cp; c < sw_n; c += KP) {
                    const int k = 16 * (sw_c0 + c) + 4 * q;
                    f32x4 x = zero4();
                    if (k < a.ldS) x = *reinterpret_cast<const f32x4*>(ds + k);
                    mfma_chunk(acc0, acc1, lds[wbs + c * 64 + lane], x);
                }
                frag[wave * 64 + lane] = acc0 + acc1;
            }
            __syncthreads();
            if (sw_n > 0 && cp == 0 && rb_on) {
                f32x4 sum = frag[wave * 64 + lane];
                for (int p = 1; p < KP; ++p) sum += frag[(rb + RB * p) * 64 + lane];
                stx(rs, sw_off + ((sw_tile * NKR + sw_kr) * 4 + q) * 1024 + (rb * 16 + j) * 16, sum);
            }
            gb::arrive(bar);
            if (wave == kWaves - 1 && red_cnt > 0)          // layer 0's ahead product of token 0
                reduce_ahead<NS>(rs, p0x_off, (H / 4) * 1024, r_off, red_o0, red_cnt, ln);
            ok = gb::wait(bar);
            if (ok) ok = gb::sync(bar);
        }
        if (ok && cellwg && cg == 0) {
            (void)t;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                f32x4 sum = l == 2 ? dhc2 : (l == 1 ? dhc1 : dhc0);
                sum += ldx(rs, r_off + l * r_bytes + ((8 * sl + wave) * kRows + row) * 16);
                if (l == 2 && direct) {
                    const f32x4* dw = lds + wbd + wave * Sd;
                    const int base = min(row, B - 1) * a.ldD * 4;
                    for (int i = 0; i < nq_dir; ++i) {
                        const f32x4 x = ldx(rd, base + 16 * i);
                        sum += x.x * dw[4 * i] + x.y * dw[4 * i + 1];
                        sum += x.z * dw[4 * i + 2] + x.w * dw[4 * i + 3];
                    }
                } else if (l == 2) {
                    const int sb = sw_off + (((2 * sl + (wave >> 2)) * NKR) * 4 + (wave & 3)) * 1024 + row * 16;
                    for (int i = 0; i < NKR; ++i) sum += ldx(rs, sb + i * 4096);
                }
                if (rvalid) {
                    *reinterpret_cast<f32x4*>(a.dh_out + (int64_t)l * BH + (int64_t)row * H + u0) = sum;
                    *reinterpret_cast<f32x4*>(a.dc_out + (int64_t)l * BH + (int64_t)row * H + u0) = l == 2 ? dc2 : (l == 1 ? dc1 : dc0);
                }
            }
        }
    }
    if (TRACE && tid == 0) stamps[15] = (unsigned)wall_clock64();      // (slot 15 of token 0: the initial-state gradients are stored)
    if (TRACE) {
        __syncthreads();
        for (int i = tid; i < a.T * kTraceSlots; i += kThreads) a.trace[(size_t)blockIdx.x * a.T * kTraceSlots + i] = stamps[i];
    }
    if (!ok) {
        gb::report_failure(a.sticky_fail, a.host_fail);
        if (tid == 0) a.poison[0] = __builtin_nanf("");
    }
}
#undef MMQG_GSTAMP

inline int64_t align_up(int64_t v, int64_t al) { return (v + al - 1) / al * al; }

// everything the kernel's geometry follows from (shape + grid size)
struct Plan {
    int NS, GS, RB, tpcL2, tpcL1, tpcL0, tpcA, nchS, NKR, cpk, cpq, n_t, n_v, n_a, lds_bytes, direct;
    int64_t szP, szP0x, szD0, szSW, szR, tok_stride;
};

bool make_plan(int B, int H, int Cw, int ldS, const mmqg_attn_values& v, int G, Plan& p) {
    const int Sd = (v.Lt + 2 * v.Lav + 3) & ~3;
    if (H % kU || H < 128 || H / kU > kMaxNS || Cw % 16 || B < 1 || B > kRows || B > G) return false;
    p.NS = H / kU;
    if (p.NS != 4 && p.NS != 8 && p.NS != 16) return false;       // the instantiated slice counts: H = 128, 256, 512
    p.GS = G / p.NS;
    if (p.GS < 2) return false;
    p.GS &= ~1;                                          // (column tile, k-range) pairs of the score-gradient product
    p.RB = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    p.tpcL2 = p.tpcL1 = ceil_div(H / 16, p.GS);
    p.tpcL0 = ceil_div(Cw / 16, p.GS);
    p.tpcA = ceil_div(H / 16, p.GS);
    p.nchS = ceil_div(ldS, 16);
    p.NKR = std::min(std::min(p.GS / 2, p.nchS), kMaxKR);
    p.cpk = ceil_div(p.nchS, p.NKR);
    p.NKR = ceil_div(p.nchS, p.cpk);
    // Short score rows: every cell lane of the top layer multiplies its row of dS(t+1) itself, which takes the SW stage
    // and its barrier out of the token (MMQG_PDB_SW_STAGE=1 keeps the stage: the A/B switch).
    static const bool keep_sw = [] { const char* e = getenv("MMQG_PDB_SW_STAGE"); return e && atoi(e) != 0; }();
    const int w_f4 = (p.tpcL2 + p.tpcL1 + p.tpcL0 + 3 * p.tpcA) * 512;
    const int lds_direct = (w_f4 + kWaves * Sd + kFragF4) * 16;
    p.direct = (!keep_sw && Sd <= 4 * kDirQ && Sd <= ldS && lds_direct <= kLdsBudget) ? 1 : 0;
    p.lds_bytes = p.direct ? lds_direct : (w_f4 + p.cpk * 64 + kFragF4) * 16;
    if (p.lds_bytes > kLdsBudget) return false;
    // attention: cpq workgroups (7 worker waves each) per question, shared out over the modalities by bytes per wave
    p.cpq = G / B;
    const int share = 7 * p.cpq;
    const int64_t rows[3] = {v.Lt, v.Lav, v.Lav}, width[3] = {v.H, v.Dv, v.Da};      // text, video, audio
    int n[3] = {1, 1, 1};
    for (int used = 3; used < share; ++used) {
        int best = -1;
        double worst = 0.0;
        for (int m = 0; m < 3; ++m) {
            if (n[m] * 4 >= rows[m]) continue;            // at least four rows (one batch) per wave
            const double load = (double)ceil_div64(rows[m], n[m]) * (double)(width[m] + 64);      // (+ per-row overhead)
            if (load > worst) { worst = load; best = m; }
        }
        if (best < 0) break;
        n[best] += 1;
    }
    p.n_t = n[0]; p.n_v = n[1]; p.n_a = n[2];
    const int pq = 2 * H / 4;
    p.szP = (int64_t)p.NS * pq * 1024;
    p.szP0x = (int64_t)p.NS * (H / 4) * 1024;
    p.szD0 = (int64_t)p.NS * kRows * Cw * 4;
    p.szSW = (int64_t)(H / 16) * p.NKR * 4 * 1024;
    p.szR = (int64_t)H * 256;
    p.tok_stride = 2 * p.szP + p.szP0x + p.szD0 + p.szSW + 3 * p.szR;
    return true;
}

struct WsLayout { int64_t bar, ex, sticky, total; };
WsLayout ws_layout(int T, const Plan& p) {
    WsLayout w;
    w.bar = 0;
    w.ex = align_up((int64_t)sizeof(gb::XBar), 256);
    w.sticky = align_up(w.ex + (int64_t)T * p.tok_stride, 256);
    w.total = w.sticky + 256;
    return w;
}

bool env_off() {
    static const bool off = [] {
        const char* e = getenv("MMQG_NO_PERSIST");
        const char* f = getenv("MMQG_NO_PERSIST_DEC_BWD");
        return (e && atoi(e) != 0) || (f && atoi(f) != 0);
    }();
    return off;
}

bool shape_ok(const mmqg_decoder_seq& d, int ld_ds) {
    if (env_off()) return false;
    const mmqg_attn_values& v = d.values;
    if (d.L != 3 || d.T < 2 || v.H != d.H) return false;
    if (v.H % 4 || v.Da % 4 || v.Dv % 4 || std::max(v.H, std::max(v.Da, v.Dv)) > kMaxD) return false;
    if (d.ld_attn % 4 || ld_ds % 4 || ld_ds < d.ld_attn) return false;
    return true;
}

}  // namespace

namespace mmqg {

static int g_dec_bwd_launches = 0;
int decoder_persist_bwd_launch_count() { return g_dec_bwd_launches; }
static unsigned long long* g_gtrace_buf = nullptr;
static int64_t g_gtrace_words = 0;
void decoder_persist_bwd_set_trace(unsigned long long* buf, int64_t words) { g_gtrace_buf = buf; g_gtrace_words = buf ? words : 0; }

int64_t decoder_persist_bwd_ws_bytes(const mmqg_decoder_seq& d, int ld_ds) {
    if (!shape_ok(d, ld_ds)) return 0;
    persist_runtime_prepare();
    const mmqg_attn_values& v = d.values;
    Plan p;
    const int cus = persist_device_cus();
    if (!make_plan(d.B, d.H, v.H + v.Da + v.Dv, d.ld_attn, v, std::min(cus > 0 ? cus : 256, 256), p)) return 0;
    const WsLayout wl = ws_layout(d.T, p);
    if (wl.sticky - wl.ex >= ((int64_t)1 << 31) - 4096) return 0;
    return wl.total;
}

// 0 = done (the whole backward time loop incl. the initial-state gradients), 1 = not taken, < 0 = error
int decoder_seq_bwd_persistent(const mmqg_decoder_seq& d, const mmqg_decoder_seq_grad& g, hipStream_t s) {
    if (!g.persist_ws || !shape_ok(d, g.ld_ds)) return 1;
    const mmqg_attn_values& v = d.values;
    const int T = d.T, B = d.B, H = d.H, Cw = v.H + v.Da + v.Dv, S = v.Lt + 2 * v.Lav;
    // every workgroup holds resident weights: the grid cannot shrink (never beside a collective: no gradient bucket is
    // final before the decoder's backward loop ends)
    const int G = std::min(persist_usable_cus(s, false), 256);
    Plan p;
    if (!make_plan(B, H, Cw, d.ld_attn, v, G, p)) return 1;
    const WsLayout wl = ws_layout(T, p);
    if (wl.sticky - wl.ex >= ((int64_t)1 << 31) - 4096) return 1;
    if (g.persist_ws_bytes < wl.total || !aligned16(g.persist_ws)) return 1;
    const float* ptrs[] = {d.w_hhT[0], d.w_hhT[1], d.w_hhT[2], d.w_ihT[1], d.w_ihT[2], d.w_ih0cT, d.w_attn_hT, d.gates, d.cs,
                           d.attn, d.ctx, g.dhtop, g.dgates, g.dscores, g.dctx, g.dh, g.dc, v.text, v.audio, v.video};
    for (const float* q : ptrs) if (!q || !aligned16(q)) return 1;
    if (v.text_stride_b % 4 || v.audio_stride_b % 4 || v.video_stride_b % 4) return 1;
    if ((v.mask_mode == MMQG_MASK_INTENDED || v.zero_past_len) && !(v.text_len && v.av_len)) return 1;
    // one instantiation per slice count (the slice loops are compile-time: a run-time bound made hipcc branch around every load)
    typedef void (*KernelFn)(DbArgs);
    const KernelFn fns[6][2] = {{decoder_persist_bwd_kernel<false, 4, false>, decoder_persist_bwd_kernel<true, 4, false>},
                                {decoder_persist_bwd_kernel<false, 8, false>, decoder_persist_bwd_kernel<true, 8, false>},
                                {decoder_persist_bwd_kernel<false, 16, false>, decoder_persist_bwd_kernel<true, 16, false>},
                                {decoder_persist_bwd_kernel<false, 4, true>, decoder_persist_bwd_kernel<true, 4, true>},
                                {decoder_persist_bwd_kernel<false, 8, true>, decoder_persist_bwd_kernel<true, 8, true>},
                                {decoder_persist_bwd_kernel<false, 16, true>, decoder_persist_bwd_kernel<true, 16, true>}};
    const int ns_i = p.NS == 4 ? 0 : (p.NS == 8 ? 1 : (p.NS == 16 ? 2 : -1));
    if (ns_i < 0) return 1;
    const int ki = ns_i + (p.direct ? 3 : 0);
    static int attr_set = 0;
    if (attr_set == 0) {
        hipError_t e = hipSuccess;
        for (int i = 0; i < 12 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[i / 2][i % 2]), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e != hipSuccess) (void)hipGetLastError();
        attr_set = e == hipSuccess ? 1 : -1;
    }
    if (attr_set < 0) return 1;
    {
        static int occ_lds = -1, occ_ki = -1, occ_ok = 0;
        if (occ_lds != p.lds_bytes || occ_ki != ki) {
            int nb = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fns[ki][0]), kThreads,
                                                                              (size_t)p.lds_bytes);
            if (e != hipSuccess) (void)hipGetLastError();
            occ_lds = p.lds_bytes; occ_ki = ki; occ_ok = (e == hipSuccess && nb >= 1) ? 1 : 0;
        }
        if (!occ_ok) return 1;
    }
    if (persist_begin(s) != 0) return 1;

    char* ws = reinterpret_cast<char*>(g.persist_ws);
    // the barrier block starts from zero.  The exchange needs no initial value: every slot is written before it is read
    // (row blocks past the batch are never written and only feed lanes whose results are dropped).
    const CopySeg init{reinterpret_cast<float*>(ws), nullptr, wl.ex / 4};
    MMQG_TRY(copy_or_zero_multi(&init, 1, s));
    const bool drop = d.training && d.dropout_p > 0.f;
    DbArgs a{};
    a.T = T; a.B = B; a.H = H; a.Cw = Cw; a.S = S; a.ldS = d.ld_attn; a.ldD = g.ld_ds; a.G = G;
    a.NS = p.NS; a.GS = p.GS; a.RB = p.RB;
    a.tpcL2 = p.tpcL2; a.tpcL1 = p.tpcL1; a.tpcL0 = p.tpcL0; a.tpcA = p.tpcA;
    a.nchS = p.nchS; a.NKR = p.NKR; a.cpk = p.cpk; a.direct = p.direct;
    a.cpq = p.cpq; a.n_t = p.n_t; a.n_v = p.n_v; a.n_a = p.n_a;
    { static const int dbg = [] { const char* e = getenv("MMQG_PDB_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
    a.v = v;
    a.w_ihT1 = d.w_ihT[1]; a.w_ihT2 = d.w_ihT[2];
    a.w_hhT0 = d.w_hhT[0]; a.w_hhT1 = d.w_hhT[1]; a.w_hhT2 = d.w_hhT[2];
    a.w_ih0cT = d.w_ih0cT; a.w_attn_hT = d.w_attn_hT;
    a.gates = d.gates; a.cs = d.cs; a.attn = d.attn; a.ctx = d.ctx; a.dhtop = g.dhtop; a.lens = d.lens;
    a.dgates = g.dgates; a.dscores = g.dscores; a.dctx = g.dctx; a.dh_out = g.dh; a.dc_out = g.dc;
    a.drop = drop ? 1 : 0; a.drop_p = drop ? d.dropout_p : 0.f; a.seed = d.seed; a.stream_base = d.stream_base; a.seed_off = d.seed_offset;
    a.ex = reinterpret_cast<float*>(ws + wl.ex);
    a.ex_bytes = (int)(wl.sticky - wl.ex);
    a.tok_stride = (int)p.tok_stride;
    a.p1_off = (int)p.szP; a.p0x_off = (int)(2 * p.szP); a.d0_off = (int)(2 * p.szP + p.szP0x); a.sw_off = (int)(2 * p.szP + p.szP0x + p.szD0);
    a.r_off = (int)(2 * p.szP + p.szP0x + p.szD0 + p.szSW);
    a.bar = reinterpret_cast<gb::XBar*>(ws + wl.bar);
    a.poison = g.dgates;
    a.sticky_fail = reinterpret_cast<unsigned*>(ws + wl.sticky);
    a.host_fail = persist_host_fail_word();
    a.expect_wg = (unsigned)(G + persist_test_extra_wg());
    a.max_spins = persist_test_max_spins() ? persist_test_max_spins() : gb::kDefaultSpins;
    a.trace = nullptr;
    const int trace_lds = T * kTraceSlots * 4;
    if (g_gtrace_buf && (int64_t)G * T * kTraceSlots <= g_gtrace_words && p.lds_bytes + trace_lds <= kLdsBudget) a.trace = g_gtrace_buf;
    if (a.dbg & 4) {
        static bool said = false;
        if (!said) fprintf(stderr, "[mmqg] decoder bwd plan: G %d NS %d GS %d RB %d tiles %d %d %d +%d, direct %d, lds %d (+%d stamps), kernel %d/%d\n",
                           G, p.NS, p.GS, p.RB, p.tpcL2, p.tpcL1, p.tpcL0, p.tpcA, p.direct, p.lds_bytes, a.trace ? trace_lds : 0, ki, a.trace ? 1 : 0);
        said = true;
    }
    hipLaunchKernelGGL(fns[ki][a.trace ? 1 : 0], dim3(G), dim3(kThreads), (size_t)(p.lds_bytes + (a.trace ? trace_lds : 0)), s, a);
    g_dec_bwd_launches += 1;
    persist_end(s);
    return check_launch("decoder_persist_bwd");
}

}  // namespace mmqg
