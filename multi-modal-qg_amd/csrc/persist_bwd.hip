// Persistent BACKWARD time loop of a stacked LSTM (autograd of nn.LSTM inside TextEncoder, model/encoder.py:91,
// as train.py:177's loss.backward() runs it): ONE launch runs all T + L - 1 anti-diagonals of the (layer, time)
// wavefront instead of one launch per diagonal (sequence.hip, lstm_seq_bwd_wavefront).
//
// A backward layer-step is
//     dh_l(t) = dG_l(t+1) W_hh_l  +  mask_l(t) * (dG_{l+1}(t) W_ih_{l+1})  +  carry / dy(t)      [B x H]
//     dG_l(t), dc_l(t-1) = cell_backward(dh_l(t), dc_l(t), saved gates and cell states)             [B x 4H]
// i.e. products with K = 4H per matrix and only H output columns: an output-stationary split would need 16 columns x
// 4096 k = 256 KB of weights per tile (LDS: 160 KB) and would leave most CUs idle.  So the weights are cut along K as
// well: the 2L - 1 recurrent matrices form (product, 64-column group) "units" of 4H/16 k-chunks each, the chunks of all
// units form one list, and every workgroup (one per CU) keeps an equal share of that list — a [k-range] x [64 columns]
// block of one or two units — in LDS in MFMA-fragment order for the whole sequence (config 2: 20 chunks = 80 KB per CU;
// the 20 MB of recurrent weights are read from memory once).  Per diagonal:
//   phase A  every workgroup multiplies its block with the matching k-range of the gate gradients of the previous
//            diagonal (exchange buffer, laid out [k/4][64 rows][4] so an operand fragment is one 16-byte load per lane)
//            on v_mfma_f32_16x16x4_f32 and stores its PARTIAL [64 rows x 64 columns] tile(s) write-through;
//   barrier  (grid_barrier.h, fence-free: every exchanged byte is stored and loaded sc1)
//   phase B  one wave per (layer, 16 hidden units, 16 rows) sums the k-slices' partial tiles (6-8 per product), applies
//            the dropout mask to the product that came through the layer above, adds carry / dy, runs the cell backward
//            for 4 units x 4 gates per lane, and publishes dG_l(t) in the exchange layout (16-byte write-through stores);
//            dc and the pass-through part of dh stay in that lane's registers for the whole sequence;
//   barrier
// dG goes to global memory (for the hoisted weight-gradient GEMMs) after the workgroup's arrival at the second
// barrier, off the critical path.  Bytes per CU and diagonal: 2 x 80 KB of operands + 16-32 KB of partial tiles,
// against 320 KB of operands for the 16-column split the forward kernel uses — the products run at the fp32 MFMA rate.
#include <stdlib.h>

#include <algorithm>

#include "grid_barrier.h"
#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

using namespace mmqg;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxL = 3;
constexpr int kMaxWG = 256;
constexpr int kThreads = 512;          // 8 waves: wave w multiplies row block (w & 3) with column tiles 2*(w >> 2), +1
constexpr int kWaves = kThreads / 64;
constexpr int kRows = 64;              // rows of the exchange layout (B <= 64)
constexpr int kColsPerUnit = 64;       // output columns of a unit
constexpr int kRing = 4;               // operand chunks (1 KB per wave) in flight; segment lengths are multiples of it
constexpr int kLdsBudget = 160 * 1024 - 1024;
constexpr int kMaxSlices = 8;          // partial tiles a phase-B wave sums per product

constexpr int kMaxCL = kMaxL + 1;      // cell layers of one launch: the stack's L layers + one layer of a second stack

// one cell layer of the launch: layer l of the stack, or the single layer of the second ("aux") stack that rides
// along (the frame LSTM's backward beside the text encoder's: its T_aux steps sit on the first T_aux diagonals)
struct CellLayer {
    const float* w_hh; const float* w_hhT;      // [4H][H] as torch keeps it / optional k-major copy [H][4H] (faster prologue)
    const float* gates; const float* cs;        // saved activations of this layer: [T][B][4H], [T+1][B][H]
    const int32_t* lens;
    const float* dy; int64_t dy_stride_t, dy_stride_b;     // gradient of the layer's outputs (top layers only), nullable
    const float* dhT; const float* dcT;         // gradient of the final state [B][H], nullable
    float* dgates;                              // out [T][B][4H]
    float* dh_out; float* dc_out;               // out [B][H]: what is left of dh / dc after step 0
    int T, doff;                                // steps; diagonal on which its step T - 1 sits
};

struct BwdArgs {
    int B, L, NL, H, G, Cper, ndiag;   // L layers of the stack, NL cell layers in all; Cper: k-chunks per workgroup
    CellLayer lay[kMaxCL];
    const float* w_ih[kMaxL]; const float* w_ihT[kMaxL];      // [4H][H] / [H][4H], layers >= 1 of the stack
    float drop_p; int drop; uint64_t seed, stream_base; const int32_t* seed_off; int T0;   // dropout streams: stream_base + l * T0 + t
    float* dgx;            // [NL][2][4H/4][64][4]  dG_l(t), slot = t & 1; directly followed by
    float* part;           // [G][2][4][64][16]    partial tiles: (workgroup, segment, column tile, row, 16 columns)
    gb::XBar* bar;
    float* poison; unsigned* sticky_fail; unsigned* host_fail; unsigned expect_wg, max_spins;
    unsigned long long* trace;
};

__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const int32_t* off) {
    return off ? seed + (uint64_t)(uint32_t)off[0] * 0x9E3779B97F4A7C15ull : seed;
}

template <typename P>
__device__ __forceinline__ P pick(P const (&arr)[kMaxL], int l) {
    static_assert(kMaxL == 3, "pick() lists three layers");
    return l == 0 ? arr[0] : (l == 1 ? arr[1] : arr[2]);
}
// field of cell layer l without indexing the by-value argument array at run time (that would make a scratch copy)
#define LSEL(field, l) ((l) == 0 ? a.lay[0].field : ((l) == 1 ? a.lay[1].field : ((l) == 2 ? a.lay[2].field : a.lay[3].field)))

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

__device__ __forceinline__ f32x4 tanh4(const f32x4& v) { return f32x4{tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)}; }

struct Seg {             // a workgroup's share of one unit
    int n;               // k-chunks (0 = none), a multiple of kRing
    int lo;              // first chunk inside the unit
    int prod, ng;        // product (0..NL-1: W_hh of cell layer prod; NL..NL+L-2: W_ih of layer prod-NL+1) and column group
    int wbase;           // LDS float4 index of its first weight fragment
    int lc, hh, lsrc, Tc, doff;   // cell layer the product feeds, W_hh or W_ih, layer whose dG it multiplies, that cell's T / doff
};

#define MMQG_BSTAMP(slot)                                                                              \
    if (TRACE && tid == 0) a.trace[((size_t)blockIdx.x * a.ndiag + s) * 6 + (slot)] = wall_clock64();

// 16 k of one chunk for the wave's two column tiles (two independent accumulator chains)
__device__ __forceinline__ void mfma_chunk2(f32x4& acc0, f32x4& acc1, const f32x4& w0, const f32x4& w1, const f32x4& x) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x.w, acc1, 0, 0, 0);
}

template <typename Rsrc>
__device__ __forceinline__ f32x4 ld_x(const Rsrc& rs, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));     // sc1: bypasses L1
}

// products of one segment for this wave: n chunks (a multiple of kRing, > 0), operand chunk i at byte offset
// xoff + i * chunk_bytes, weight fragments of chunk i / column tile ct at lds[wbase + (i * 4 + ct) * 64 + lane].
// No branch inside the loops (hipcc then counts the outstanding loads exactly instead of draining them).
template <typename Rsrc>
__device__ __forceinline__ void seg_products(f32x4 (&ring)[kRing], const Rsrc& rs, int xoff, int n, const f32x4* wl, f32x4& acc0,
                                             f32x4& acc1) {
    constexpr int chunk_bytes = 4 * kRows * 16;
    for (int i0 = 0; i0 + kRing < n; i0 += kRing) {
#pragma unroll
        for (int d = 0; d < kRing; ++d) {
            const f32x4 w0 = wl[(i0 + d) * 256], w1 = wl[(i0 + d) * 256 + 64];
            mfma_chunk2(acc0, acc1, w0, w1, ring[d]);
            ring[d] = ld_x(rs, xoff + (i0 + kRing + d) * chunk_bytes);
        }
    }
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
        const f32x4 w0 = wl[(n - kRing + d) * 256], w1 = wl[(n - kRing + d) * 256 + 64];
        mfma_chunk2(acc0, acc1, w0, w1, ring[d]);
    }
}

// the same for ONE column tile per wave (batches of at most 32 rows: two row blocks x four column tiles over the eight
// waves instead of four row blocks x two tiles); two accumulator chains over alternating k (the 16x16x4 f32 MFMA has 40
// cycles of dependent latency against 32 of issue)
template <typename Rsrc>
__device__ __forceinline__ void seg_products1(f32x4 (&ring)[kRing], const Rsrc& rs, int xoff, int n, const f32x4* wl, f32x4& acc) {
    constexpr int chunk_bytes = 4 * kRows * 16;
    f32x4 a0 = acc, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 + kRing < n; i0 += kRing) {
#pragma unroll
        for (int d = 0; d < kRing; ++d) {
            const f32x4 w = wl[(i0 + d) * 256], x = ring[d];
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, a1, 0, 0, 0);
            ring[d] = ld_x(rs, xoff + (i0 + kRing + d) * chunk_bytes);
        }
    }
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
        const f32x4 w = wl[(n - kRing + d) * 256], x = ring[d];
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, a1, 0, 0, 0);
    }
    acc = a0 + a1;
}

// NARROW: B <= 32 (config 4): only two 16-row blocks hold questions
template <bool TRACE, bool NARROW>
__global__ __launch_bounds__(kThreads, 2) void lstm_persist_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    gb::Ctx bar;
    gb::init_arrive(bar, a.bar, a.max_spins);     // the census / flat barrier turns while the weights are on their way to the LDS
    const int B = a.B, L = a.L, NL = a.NL, H = a.H;
    const int CPU = H / 4;                       // k-chunks per unit (K = 4H, 16 k per chunk)
    const int NG = H / kColsPerUnit;             // column groups per product
    const int nunits = (NL + L - 1) * NG;
    const int total = nunits * CPU;
    const int slot_f = kRows * 4 * H;            // floats per (layer, slot) of the dG exchange buffer

    // ---- this workgroup's share of the chunk list: one or two segments
    Seg sg[2];
    {
        const int start = blockIdx.x * a.Cper;
        const int count = max(0, min(a.Cper, total - start));
        const int u0 = start / CPU, lo0 = start - u0 * CPU;
        sg[0].n = min(CPU - lo0, count); sg[0].lo = lo0; sg[0].prod = u0 / NG; sg[0].ng = u0 - sg[0].prod * NG; sg[0].wbase = 0;
        const int u1 = u0 + 1;
        sg[1].n = count - sg[0].n; sg[1].lo = 0; sg[1].prod = u1 / NG; sg[1].ng = u1 - sg[1].prod * NG; sg[1].wbase = sg[0].n * 256;
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            Seg& g = sg[si];
            g.hh = g.prod < NL ? 1 : 0;
            g.lc = g.hh ? g.prod : g.prod - NL;                      // W_ih of layer lc + 1 feeds cell layer lc
            g.lsrc = g.hh ? g.lc : g.lc + 1;
            g.Tc = LSEL(T, g.lc); g.doff = LSEL(doff, g.lc);
        }
    }
    // ---- weights -> LDS in fragment order: chunk c, column tile ct, lane (i = lane & 15: output column, kq = lane >> 4):
    // W[k = 16 c + 4 kq + {0..3}][n = 64 ng + 16 ct + i] of the product's matrix.  From the k-major copy [H][4H] when the
    // caller keeps one (one 16-byte load), else from the [4H][H] original (four loads H apart).
#pragma unroll
    for (int si = 0; si < 2; ++si) {
        const Seg& g = sg[si];
        const float* W = g.hh ? LSEL(w_hh, g.lc) : pick(a.w_ih, g.lsrc);
        const float* WT = g.hh ? LSEL(w_hhT, g.lc) : pick(a.w_ihT, g.lsrc);
        const int nf = g.n * 256;
        for (int idx = tid; idx < nf; idx += kThreads) {
            const int c = idx >> 8, ct = (idx >> 6) & 3, l = idx & 63, i = l & 15, kq = l >> 4;
            const int k = 16 * (g.lo + c) + 4 * kq, n = kColsPerUnit * g.ng + 16 * ct + i;
            if (WT) {
                lds[g.wbase + idx] = *reinterpret_cast<const f32x4*>(WT + (int64_t)n * 4 * H + k);
            } else {
                const float* src = W + (int64_t)k * H + n;
                lds[g.wbase + idx] = f32x4{src[0], src[H], src[2 * H], src[3 * H]};
            }
        }
    }

    // one descriptor over the dG exchange buffer and the partial tiles behind it (sc1 loads / stores: aux = 16)
    const int part_base = NL * 2 * slot_f * 4;                        // byte offset of the partial tiles
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.dgx, 0, part_base + a.G * 2 * 4 * kRows * 16 * 4, 0x00020000);

    // ---- phase-A role of this wave: row block rb, column tiles 2 ch and 2 ch + 1
    const int rb = NARROW ? (wave & 1) : (wave & 3), ch = NARROW ? (wave >> 1) : (wave >> 2);     // NARROW: ch = the wave's ONE column tile
    const int j = lane & 15, q = lane >> 4;
    const int xlane = (q * kRows + rb * 16 + j) * 16;                 // byte offset inside an operand chunk
    const int plane = ((rb * 16 + j) * 16 + 4 * q) * 4;               // byte offset inside a [64][16] partial column tile

    // ---- phase-B role: wave-task tau = (cell layer, 16 hidden units, 16 rows); lane = (row r, unit quad cq)
    const int per_layer = (H / 16) * 4;
    const int ntask = NL * per_layer;
    const int tau = blockIdx.x + a.G * wave;
    const bool has_task = tau < ntask && (tau & 3) * 16 < B;           // (row blocks past the batch: nothing to do, their exchange rows stay zero)
    const int tl = has_task ? tau / per_layer : 0;                    // cell layer
    const int tct = has_task ? (tau >> 2) % (H / 16) : 0;             // 16-unit tile
    const int trb = tau & 3;                                          // row block
    const int br = trb * 16 + (lane >> 2), cq = lane & 3;
    const int u0 = tct * 16 + 4 * cq;                                 // first of this lane's 4 hidden units
    const bool bvalid = has_task && br < B;
    // this wave's layer, once (wave-uniform values)
    const int Tl = LSEL(T, tl), doffl = LSEL(doff, tl);
    const float* gates_l = LSEL(gates, tl);
    const float* cs_l = LSEL(cs, tl);
    const float* dy_l = LSEL(dy, tl);
    const int64_t dy_st = LSEL(dy_stride_t, tl), dy_sb = LSEL(dy_stride_b, tl);
    float* dgates_l = LSEL(dgates, tl);
    const bool stack_layer = tl < L;                                  // (else: the second stack's layer)
    const bool has_above = stack_layer && tl < L - 1;
    int blen = Tl;
    f32x4 dhc = f32x4{0.f, 0.f, 0.f, 0.f}, dcc = dhc;                 // pass-through dh and dc of (tl, br, u0..u0+3)
    if (bvalid) {
        const int32_t* lens = LSEL(lens, tl);
        const float* dhT = LSEL(dhT, tl);
        const float* dcT = LSEL(dcT, tl);
        if (lens) blen = lens[br];
        if (dhT) dhc = *reinterpret_cast<const f32x4*>(dhT + (int64_t)br * H + u0);
        if (dcT) dcc = *reinterpret_cast<const f32x4*>(dcT + (int64_t)br * H + u0);
    }
    // where the partial tiles of this task's two products come from: unit -> workgroups [cA, cB], segment index
    const int tng = tct >> 2, tctl = tct & 3;
    int hhA, hhN, ihA, ihN, hh_first, ih_first;
    {
        const int uh = tl * NG + tng, ui = (NL + tl) * NG + tng;
        hhA = (uh * CPU) / a.Cper; hhN = ((uh + 1) * CPU - 1) / a.Cper - hhA + 1; hh_first = uh * CPU;
        ihA = (ui * CPU) / a.Cper; ihN = ((ui + 1) * CPU - 1) / a.Cper - ihA + 1; ih_first = ui * CPU;
    }
    const int pb_lane = ((tctl * kRows + br) * 16 + 4 * cq) * 4;       // byte offset inside a (workgroup, segment) partial block

    // (announced at kernel entry, before the weights were read; no second barrier: the first diagonal reads nothing that
    // another workgroup of this launch has written and ends with an arrival of its own)
    bool ok = gb::init_wait(bar, a.expect_wg);
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    const uint64_t seed = a.drop ? eff_seed(a.seed, a.seed_off) : 0;
    for (int s = 0; ok && s < a.ndiag; ++s) {
        MMQG_BSTAMP(0)
        // ---- phase B operands that do not depend on the chain: requested now, used after the first barrier
        const int tt = (Tl - 1) - (s - doffl);                        // time of this wave's cell on this diagonal
        const bool con = has_task && tt >= 0 && tt < Tl;
        f32x4 gi, gf, gg, go, cprev, cnew, dy4;
        gi = gf = gg = go = cprev = cnew = dy4 = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool cact = con && bvalid && tt < blen;
        if (cact) {
            const float* gr = gates_l + ((int64_t)tt * B + br) * 4 * H + u0;
            gi = *reinterpret_cast<const f32x4*>(gr); gf = *reinterpret_cast<const f32x4*>(gr + H);
            gg = *reinterpret_cast<const f32x4*>(gr + 2 * H); go = *reinterpret_cast<const f32x4*>(gr + 3 * H);
            const float* cr = cs_l + ((int64_t)tt * B + br) * H + u0;
            cprev = *reinterpret_cast<const f32x4*>(cr); cnew = *reinterpret_cast<const f32x4*>(cr + (int64_t)B * H);
            if (dy_l) dy4 = *reinterpret_cast<const f32x4*>(dy_l + (int64_t)tt * dy_st + (int64_t)br * dy_sb + u0);
        }

        // ---- phase A: partial products of this workgroup's segments
        f32x4 ring0[kRing], ring1[kRing];
        int xoff[2]; bool son[2];
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            const Seg& g = sg[si];
            const int tc = (g.Tc - 1) - (s - g.doff);
            // W_hh_l needs dG_l(t+1) (slot (t+1)&1 of layer l); W_ih_{l+1} needs dG_{l+1}(t) (slot t&1 of layer l+1)
            son[si] = g.n > 0 && tc >= 0 && tc < g.Tc && (!g.hh || tc + 1 < g.Tc);
            const int slot = g.hh ? ((tc + 1) & 1) : (tc & 1);
            xoff[si] = ((g.lsrc * 2 + slot) * slot_f) * 4 + (4 * g.lo) * kRows * 16 + xlane;
        }
        constexpr int chunk_bytes = 4 * kRows * 16;
        if (son[0]) {
#pragma unroll
            for (int d = 0; d < kRing; ++d) ring0[d] = ld_x(rs, xoff[0] + d * chunk_bytes);
        }
        if (son[1]) {
#pragma unroll
            for (int d = 0; d < kRing; ++d) ring1[d] = ld_x(rs, xoff[1] + d * chunk_bytes);
        }
        if (NARROW) {
            if (son[0]) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                seg_products1(ring0, rs, xoff[0], sg[0].n, lds + sg[0].wbase + ch * 64 + lane, acc);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rs,
                                                       part_base + ((blockIdx.x * 2 + 0) * 4 + ch) * kRows * 16 * 4 + plane, 0, 16);
            }
            if (son[1]) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                seg_products1(ring1, rs, xoff[1], sg[1].n, lds + sg[1].wbase + ch * 64 + lane, acc);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rs,
                                                       part_base + ((blockIdx.x * 2 + 1) * 4 + ch) * kRows * 16 * 4 + plane, 0, 16);
            }
        } else {
        if (son[0]) {
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            seg_products(ring0, rs, xoff[0], sg[0].n, lds + sg[0].wbase + (2 * ch) * 64 + lane, acc0, acc1);
            const int pb = part_base + ((blockIdx.x * 2 + 0) * 4 + 2 * ch) * kRows * 16 * 4 + plane;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc0), rs, pb, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc1), rs, pb + kRows * 16 * 4, 0, 16);
        }
        if (son[1]) {
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            seg_products(ring1, rs, xoff[1], sg[1].n, lds + sg[1].wbase + (2 * ch) * 64 + lane, acc0, acc1);
            const int pb = part_base + ((blockIdx.x * 2 + 1) * 4 + 2 * ch) * kRows * 16 * 4 + plane;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc0), rs, pb, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc1), rs, pb + kRows * 16 * 4, 0, 16);
        }
        }
        MMQG_BSTAMP(1)
        ok = gb::sync(bar);
        MMQG_BSTAMP(2)
        if (!ok) break;

        // ---- phase B: sum the k-slices, cell backward, publish dG_l(t)
        f32x4 dgi, dgf, dgg, dgo;
        dgi = dgf = dgg = dgo = f32x4{0.f, 0.f, 0.f, 0.f};
        if (con) {
            f32x4 ph = f32x4{0.f, 0.f, 0.f, 0.f}, pi = ph;
            if (tt + 1 < Tl) {
                f32x4 v[kMaxSlices];
#pragma unroll
                for (int i = 0; i < kMaxSlices; ++i) {
                    const int cu = hhA + min(i, hhN - 1);
                    const int seg = cu * a.Cper < hh_first ? 1 : 0;
                    v[i] = ld_x(rs, part_base + (cu * 2 + seg) * (4 * kRows * 16 * 4) + pb_lane);
                }
#pragma unroll
                for (int i = 0; i < kMaxSlices; ++i) {
                    const float w = i < hhN ? 1.f : 0.f;
                    ph.x += w * v[i].x; ph.y += w * v[i].y; ph.z += w * v[i].z; ph.w += w * v[i].w;
                }
            }
            if (has_above) {
                f32x4 v[kMaxSlices];
#pragma unroll
                for (int i = 0; i < kMaxSlices; ++i) {
                    const int cu = ihA + min(i, ihN - 1);
                    const int seg = cu * a.Cper < ih_first ? 1 : 0;
                    v[i] = ld_x(rs, part_base + (cu * 2 + seg) * (4 * kRows * 16 * 4) + pb_lane);
                }
#pragma unroll
                for (int i = 0; i < kMaxSlices; ++i) {
                    const float w = i < ihN ? 1.f : 0.f;
                    pi.x += w * v[i].x; pi.y += w * v[i].y; pi.z += w * v[i].z; pi.w += w * v[i].w;
                }
                if (a.drop) {
                    const f32x4 m = dropout_scale4(seed, a.stream_base + (uint64_t)tl * a.T0 + tt, (uint64_t)((int64_t)br * H + u0), a.drop_p);
                    pi.x *= m.x; pi.y *= m.y; pi.z *= m.z; pi.w *= m.w;
                }
            }
            if (bvalid) {
                f32x4 dh = dhc + ph + pi;
                if (cact) {
                    dh += dy4;
                    const f32x4 tc4 = tanh4(cnew);
                    const f32x4 dct = dcc + dh * go * (1.f - tc4 * tc4);
                    dgi = dct * gg * gi * (1.f - gi);
                    dgf = dct * cprev * gf * (1.f - gf);
                    dgg = dct * gi * (1.f - gg * gg);
                    dgo = dh * tc4 * go * (1.f - go);
                    dcc = dct * gf;
                    dhc = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    dhc = dh;                       // finished row: its state was carried forward, so is its gradient
                }
            }
            // exchange layout [k/4][64 rows][4] with k = gate * H + unit: one 16-byte write-through store per gate
            const int xb = ((tl * 2 + (tt & 1)) * slot_f) * 4 + ((u0 >> 2) * kRows + br) * 16;
            const int gstep = (H >> 2) * kRows * 16;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dgi), rs, xb, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dgf), rs, xb + gstep, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dgg), rs, xb + 2 * gstep, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dgo), rs, xb + 3 * gstep, 0, 16);
        }
        MMQG_BSTAMP(3)
        gb::arrive(bar);
        MMQG_BSTAMP(4)
        // ---- the gate gradients for the hoisted weight-gradient products: after the arrival, off the critical path
        if (con && bvalid) {
            float* dg = dgates_l + ((int64_t)tt * B + br) * 4 * H + u0;
            *reinterpret_cast<f32x4*>(dg) = dgi; *reinterpret_cast<f32x4*>(dg + H) = dgf;
            *reinterpret_cast<f32x4*>(dg + 2 * H) = dgg; *reinterpret_cast<f32x4*>(dg + 3 * H) = dgo;
        }
        ok = gb::wait(bar);
        MMQG_BSTAMP(5)
    }
    // what is left of dh / dc after step 0: the gradient of the initial state still lacks dG_l(0) W_hh_l (host side)
    if (bvalid) {
        float* dh_out = LSEL(dh_out, tl);
        float* dc_out = LSEL(dc_out, tl);
        if (dh_out) *reinterpret_cast<f32x4*>(dh_out + (int64_t)br * H + u0) = dhc;
        if (dc_out) *reinterpret_cast<f32x4*>(dc_out + (int64_t)br * H + u0) = dcc;
    }
    if (!ok) {
        gb::report_failure(a.sticky_fail, a.host_fail);
        if (tid == 0) a.poison[0] = __builtin_nanf("");
    }
}
#undef MMQG_BSTAMP

inline int64_t align_up(int64_t v, int64_t al) { return (v + al - 1) / al * al; }

struct WsLayout { int64_t bar, dgx, part, sticky, total; };
WsLayout ws_layout(int NL, int H, int G) {
    WsLayout w;
    w.bar = 0;
    w.dgx = align_up((int64_t)sizeof(gb::XBar), 256);
    w.part = w.dgx + (int64_t)NL * 2 * kRows * 4 * H * 4;
    w.sticky = w.part + (int64_t)G * 2 * 4 * kRows * 16 * 4;
    w.total = w.sticky + 256;
    return w;
}

// k-chunks per workgroup for G workgroups (a multiple of kRing), 0 = the shape does not fit
int chunks_per_wg(int nprod, int H, int G) {
    const int total = nprod * (H / kColsPerUnit) * (H / 4);
    int per = ceil_div(total, G);
    per = std::max(per, ceil_div(H / 4, kMaxSlices - 1));      // narrow layers: fewer, longer slices (some CUs stay idle)
    per = ceil_div(per, kRing) * kRing;
    if (per > H / 4) return 0;                      // a workgroup's share must span at most two units
    if ((int64_t)per * 4096 > kLdsBudget) return 0;
    // a unit's partial tiles come from at most kMaxSlices workgroups
    if (ceil_div(H / 4, per) + 1 > kMaxSlices) return 0;
    return per;
}

}  // namespace

namespace mmqg {

static int g_persist_bwd_launches = 0;
int persist_bwd_launch_count() { return g_persist_bwd_launches; }
static unsigned long long* g_btrace_buf = nullptr;
static int64_t g_btrace_words = 0;
void persist_bwd_set_trace(unsigned long long* buf, int64_t words) { g_btrace_buf = buf; g_btrace_words = buf ? words : 0; }

bool lstm_persist_bwd_shape_ok(int T, int B, int L, int H) {
    static const bool off = [] {
        const char* e = getenv("MMQG_NO_PERSIST");
        const char* e2 = getenv("MMQG_NO_PERSIST_BWD");
        return (e && atoi(e) != 0) || (e2 && atoi(e2) != 0);
    }();
    if (off) return false;
    return T >= 2 && B >= 1 && B <= kRows && L >= 1 && L <= kMaxL && H >= 128 && H % kColsPerUnit == 0 &&
           (H / 4) % kRing == 0 && (L + 1) * (H / 16) * 4 <= kMaxWG * kWaves;
}

int64_t lstm_persist_bwd_ws_bytes(int T, int B, int L, int H) {
    if (!lstm_persist_bwd_shape_ok(T, B, L, H)) return 0;
    persist_runtime_prepare();
    const int G = std::min(persist_device_cus(), kMaxWG);
    // room for one cell layer of a second stack riding along (mmqg_lstm_seq_bwd_pair); a CPU-only caller sizing
    // buffers gets the 256-workgroup layout
    return ws_layout(L + 1, H, G >= 64 ? G : kMaxWG).total;
}

static bool grads_aligned(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g) {
    if (!aligned16(d.gates) || !aligned16(d.cs) || !aligned16(g.dgates) || !aligned16(g.dh) || !aligned16(g.dc)) return false;
    if (g.dy && (!aligned16(g.dy) || g.dy_stride_t % 4 || g.dy_stride_b % 4)) return false;
    if ((g.dhT && !aligned16(g.dhT)) || (g.dcT && !aligned16(g.dcT))) return false;
    return true;
}

// 0 = done, 1 = not eligible (the caller takes the launch-per-diagonal path), < 0 = error.
// d2 / g2 (nullable): a second, single-layer stack of the same B and H whose backward time loop rides along in the same
// launch (its T2 steps on the first T2 diagonals).
int lstm_seq_bwd_persistent(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, const mmqg_lstm_seq* d2,
                            const mmqg_lstm_seq_grad* g2, hipStream_t s) {
    if (!g.persist_ws || !lstm_persist_bwd_shape_ok(d.T, d.B, d.L, d.H)) return 1;
    const int T = d.T, B = d.B, H = d.H, L = d.L;
    const bool aux = d2 && g2;
    if (aux && (d2->L != 1 || d2->H != H || d2->B != B || d2->T < 1 || !d2->w_hh[0] || !g2->dgates || !g2->dh || !g2->dc ||
                !grads_aligned(*d2, *g2)))
        return 1;
    const int NL = L + (aux ? 1 : 0);
    // the chunk list is cut for any grid size: in data-parallel runs this launch is the one that overlaps the first two
    // gradient buckets' all-reduces and leaves the reserved CUs to RCCL's channels (persist_set_reserved_cus)
    const int G = std::min(persist_usable_cus(s, true), kMaxWG);
    if (G < 64) return 1;
    const int per = chunks_per_wg(NL + L - 1, H, G);
    if (per == 0) return 1;
    const WsLayout wl = ws_layout(NL, H, G);
    if (g.persist_ws_bytes < wl.total || !aligned16(g.persist_ws)) return 1;
    for (int l = 0; l < L; ++l)
        if (!d.w_hh[l] || (l > 0 && !d.w_ih[l])) return 1;
    if (!grads_aligned(d, g)) return 1;
    for (int l = 0; l < L; ++l)      // the k-major copies are optional, but must be 16-byte aligned to be used
        if ((d.w_hhT[l] && !aligned16(d.w_hhT[l])) || (l > 0 && d.w_ihT[l] && !aligned16(d.w_ihT[l]))) return 1;
    const int lds_bytes = per * 4096;
    static int attr_set = 0;
    if (attr_set == 0) {
        hipError_t e = hipSuccess;
        const void* fns[4] = {reinterpret_cast<const void*>(lstm_persist_bwd_kernel<false, false>),
                              reinterpret_cast<const void*>(lstm_persist_bwd_kernel<true, false>),
                              reinterpret_cast<const void*>(lstm_persist_bwd_kernel<false, true>),
                              reinterpret_cast<const void*>(lstm_persist_bwd_kernel<true, true>)};
        for (int i = 0; i < 4 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e != hipSuccess) (void)hipGetLastError();
        attr_set = e == hipSuccess ? 1 : -1;
    }
    if (attr_set < 0) return 1;
    {
        static int occ_lds = -1, occ_ok = 0;
        if (occ_lds != lds_bytes) {
            int nb = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &nb, reinterpret_cast<const void*>(lstm_persist_bwd_kernel<false, false>), kThreads, (size_t)lds_bytes);
            if (e != hipSuccess) (void)hipGetLastError();
            occ_lds = lds_bytes; occ_ok = (e == hipSuccess && nb >= 1) ? 1 : 0;
        }
        if (!occ_ok) return 1;
    }
    if (persist_begin(s) != 0) return 1;

    char* ws = reinterpret_cast<char*>(g.persist_ws);
    // the barrier block and the dG exchange buffer start from zero (rows >= B and not-yet-written slots read as zero)
    MMQG_TRY(copy_or_zero_f32(reinterpret_cast<float*>(ws), nullptr, wl.part / 4, s));
    const int64_t BH = (int64_t)B * H, G4 = (int64_t)B * 4 * H;
    BwdArgs a{};
    a.B = B; a.L = L; a.NL = NL; a.H = H; a.G = G; a.Cper = per;
    a.ndiag = std::max(T + L - 1, aux ? d2->T : 0);
    for (int l = 0; l < L; ++l) {
        CellLayer& c = a.lay[l];
        c.w_hh = d.w_hh[l]; c.w_hhT = d.w_hhT[l];
        c.gates = d.gates + (int64_t)l * T * G4; c.cs = d.cs + (int64_t)l * (T + 1) * BH; c.lens = d.lens;
        if (l == L - 1) { c.dy = g.dy; c.dy_stride_t = g.dy_stride_t; c.dy_stride_b = g.dy_stride_b; }
        c.dhT = g.dhT ? g.dhT + l * BH : nullptr; c.dcT = g.dcT ? g.dcT + l * BH : nullptr;
        c.dgates = g.dgates + (int64_t)l * T * G4; c.dh_out = g.dh + l * BH; c.dc_out = g.dc + l * BH;
        c.T = T; c.doff = L - 1 - l;
        a.w_ih[l] = l > 0 ? d.w_ih[l] : nullptr; a.w_ihT[l] = l > 0 ? d.w_ihT[l] : nullptr;
    }
    if (aux) {
        CellLayer& c = a.lay[L];
        c.w_hh = d2->w_hh[0]; c.w_hhT = (d2->w_hhT[0] && aligned16(d2->w_hhT[0])) ? d2->w_hhT[0] : nullptr;
        c.gates = d2->gates; c.cs = d2->cs; c.lens = d2->lens;
        c.dy = g2->dy; c.dy_stride_t = g2->dy_stride_t; c.dy_stride_b = g2->dy_stride_b;
        c.dhT = g2->dhT; c.dcT = g2->dcT;
        c.dgates = g2->dgates; c.dh_out = g2->dh; c.dc_out = g2->dc;
        c.T = d2->T; c.doff = 0;
    }
    a.drop = (d.training && d.dropout_p > 0.f && L > 1) ? 1 : 0;
    a.drop_p = a.drop ? d.dropout_p : 0.f; a.seed = d.seed; a.stream_base = d.stream_base; a.seed_off = d.seed_offset; a.T0 = T;
    a.bar = reinterpret_cast<gb::XBar*>(ws + wl.bar);
    a.dgx = reinterpret_cast<float*>(ws + wl.dgx);
    a.part = reinterpret_cast<float*>(ws + wl.part);
    a.poison = g.dgates;
    a.sticky_fail = reinterpret_cast<unsigned*>(ws + wl.sticky);
    a.host_fail = persist_host_fail_word();
    a.expect_wg = (unsigned)(G + persist_test_extra_wg());
    a.max_spins = persist_test_max_spins() ? persist_test_max_spins() : gb::kDefaultSpins;
    a.trace = nullptr;
    if (g_btrace_buf && (int64_t)G * a.ndiag * 6 <= g_btrace_words) a.trace = g_btrace_buf;
    static const bool no_narrow = [] { const char* e = getenv("MMQG_PERSIST_BWD_NO_NARROW"); return e && atoi(e) != 0; }();
    const bool narrow = B <= 32 && !no_narrow;
    if (a.trace && narrow) hipLaunchKernelGGL((lstm_persist_bwd_kernel<true, true>), dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a);
    else if (a.trace) hipLaunchKernelGGL((lstm_persist_bwd_kernel<true, false>), dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a);
    else if (narrow) hipLaunchKernelGGL((lstm_persist_bwd_kernel<false, true>), dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a);
    else hipLaunchKernelGGL((lstm_persist_bwd_kernel<false, false>), dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a);
    g_persist_bwd_launches += 1;
    persist_end(s);
    return check_launch("lstm_persist_bwd");
}

}  // namespace mmqg
