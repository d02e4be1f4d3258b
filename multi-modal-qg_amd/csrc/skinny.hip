// Batch-sized ("skinny", M = questions per GPU) products of the recurrent loops, fused with the
// step's pointwise epilogue so ONE launch does a whole LSTM layer-step:
//
//   acc[b][n] = sum over up to 3 operand pairs of  A_p[b][:K_p] . B_p[n][:K_p]      (all k-major)
//
//   PLAIN     C = (beta ? C : 0) + bias + acc                      (scores += h_top W_attn_h^T, dctx, dh0)
//   FWD_CELL  n runs over gate columns; acc (+ hoisted pre-activations + biases) -> LSTM cell
//             update -> h, c, activated gates, dropped h, masked top output
//             (model/encoder.py:54,91 and model/decoder.py:69 nn.LSTM, one time step)
//   BWD_CELL  n runs over hidden units; acc = recurrent + from-above gradient of h(t) (the latter
//             through that layer's dropout mask) -> cell backward -> dgates(t), dc
//
// Why not the tiled LDS GEMM: at M = 64 a layer-step is 0.13-0.43 GFLOP spread over only
// 64x2048 (or 64x512) outputs; what bounds it is launch + memory latency, not MFMA throughput.
// So: one workgroup per 16x16 output tile, its KS wavefronts split K between them (no atomics,
// no zero-fill pass, no second kernel), operands go global -> VGPR directly as 16-byte lane
// loads issued four k-chunks ahead (GEMV-style, no LDS round trip), v_mfma_f32_16x16x4_f32
// (exact fp32) does the arithmetic, the KS partial tiles are combined through 2-4 KB of LDS and
// the epilogue runs on the tile's 256 outputs.  512 (fwd) / 128-256 (bwd) workgroups of 4-8
// waves fill the 256 CUs.  Backward products use transposed weight copies (refreshed after each
// optimizer step by transpose_f32) so that every B operand is k-major too.
#include <stdlib.h>

#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MODE_PLAIN = 0, MODE_FWD_CELL = 1, MODE_BWD_CELL = 2 };

struct Pair {
    const float* A; int lda;
    const float* B; int ldb;
    int K;
    int masked;     // BWD_CELL: multiply this pair's product by the dropout mask of (stream_id, b, j)
};

struct SkinnyK {
    int M, N;
    int plain;                     // job of a BWD_CELL launch that is a plain product (C/ldc/beta/bias epilogue)
    Pair p[3];
    int cs1, cs2, chunks;          // first chunk of pair 1 / pair 2, total 16-wide k-chunks
    // PLAIN
    float* C; int ldc; int beta; const float* bias;
    // cell (both directions)
    int H;
    const int32_t* lens; int t;
    float drop_p; uint64_t seed; uint64_t stream_id; const int32_t* seed_off;
    // FWD_CELL
    float* gates; int gates_has_pre; const float* bias1; const float* bias2;
    const float* h_prev; const float* c_prev; float* h_out; float* c_out; float* h_drop;
    float* y_out; int64_t y_stride_b;
    // BWD_CELL
    const float* gates_act; const float* c_new;   // c_prev shared with fwd field
    float* carry;                                  // [M][H] in: carried gradient of finished rows / dhT; out: carry for t-1
    const float* above; int64_t above_stride_b;    // element-wise gradient from the layer above (masked)
    const float* extra; int64_t extra_stride_b;    // further gradient of h(t), active rows only
    const float* pre;                              // [M][H] product formed ahead of time, every row
    float* dc; float* dgates;
};

__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const int32_t* off) {
    return off ? seed + (uint64_t)(uint32_t)off[0] * 0x9E3779B97F4A7C15ull : seed;
}

struct SkinnyBatch {
    SkinnyK job[3];      // blockIdx.z selects the job: independent layer-steps of one wavefront diagonal
    // wide backward / plain kernel only: k slices per tile and the workspace their partial tiles meet in
    int ksl;
    float* ws;           // [job][tile_m][tile_n][slice][2 (plain | masked)][64][32] partial tiles, then the tickets
    unsigned* tickets;   // [job][tile_m][tile_n], zero between launches
    int ws_tiles_n;      // tile_n extent of the workspace indexing (largest job of the launch)
    int ws_tiles_m;
};

// tools/skinny_probe.hip compiles this file with MMQG_SKINNY_TRACE to stamp the stages of a workgroup
#ifdef MMQG_SKINNY_TRACE
__device__ unsigned long long* g_skinny_trace = nullptr;
#define MMQG_STAMP(slot)                                                                                          \
    if (g_skinny_trace && threadIdx.x == 0)                                                                       \
        g_skinny_trace[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64();
#else
#define MMQG_STAMP(slot)
#endif

// The 4-wave forward variant serves the three-job wavefront launches (1536 workgroups): at <= 85 VGPRs six
// of its workgroups fit a CU, so the whole launch is resident at once instead of needing a second round
// (31 -> 25 us per launch).  The 8-wave backward variant is held to 128 VGPRs for two workgroups per CU.
template <int MODE, int KS>
__global__ __launch_bounds__(KS * 64, (MODE == MODE_FWD_CELL && KS == 4) ? 6 : (MODE == MODE_BWD_CELL && KS == 8) ? 4 : 1) void skinny_kernel(SkinnyBatch batch) {
    __shared__ float part[2][KS][16][17];
    MMQG_STAMP(0)
    const SkinnyK& a = batch.job[blockIdx.z];
    // a cell launch (either direction) may carry plain products as further jobs: look-ahead products of the time loops
    const bool fwd_cell = MODE == MODE_FWD_CELL && !a.plain;
    if ((int)blockIdx.y * 16 >= a.M || (int)blockIdx.x * (fwd_cell ? 4 : 16) >= (fwd_cell ? a.H : a.N))
        return;   // jobs of one launch may differ in size
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;      // MFMA operand slot of this lane: tile row/col c, k-slot kq
    const int m0 = blockIdx.y * 16;
    const int n0 = blockIdx.x * (fwd_cell ? 4 : 16);                  // FWD cell: first hidden unit of the tile
    // Global loads are issued with 4 adjacent lanes on 64 contiguous bytes of ONE row (lr = lane>>2,
    // 16-byte segment ls = lane&3) so each lane quad is one cache-line access; a lane-per-row order
    // (what the MFMA operand layout wants) made every wave load 64 separate line accesses and the
    // kernel L1-bound.  One ds_bpermute per dword then moves the data to the MFMA slots:
    // slot (c, kq) takes from loader lane 4*c + kq.
    const int lr = lane >> 2, ls = lane & 3;
    const int src_lane = 4 * c + kq;
    const int ra = min(m0 + lr, a.M - 1);
    int nb;
    if (fwd_cell) nb = (lr >> 2) * a.H + n0 + (lr & 3);               // gate block (lr>>2), unit n0 + (lr&3)
    else nb = min(n0 + lr, a.N - 1);

    const int per = (a.chunks + KS - 1) / KS;
    const int q0 = wave * per, q1 = min(a.chunks, q0 + per);

    // Epilogue operands are requested now so their latency hides behind the operand stream.
    const int e_row = (threadIdx.x >> 4) & 15, e_col = threadIdx.x & 15;
    const int e_b = m0 + e_row;
    // (no arithmetic on the fetched values here: an add would make the wave wait for them before it
    // may issue its operand loads)
    float pf0 = 0.f, pf1 = 0.f, pf2 = 0.f, pf3 = 0.f, pf4 = 0.f, pf5 = 0.f, pf6 = 0.f;
    if (threadIdx.x < 256 && e_b < a.M) {
        if (fwd_cell) {
            const int gc = (e_col >> 2) * a.H + n0 + (e_col & 3);
            if (a.gates_has_pre) pf0 = a.gates[(int64_t)e_b * 4 * a.H + gc];
            if (a.bias1) pf3 = a.bias1[gc];
            if (a.bias2) pf4 = a.bias2[gc];
            if ((e_col >> 2) == 0) {
                const int64_t e = (int64_t)e_b * a.H + n0 + (e_col & 3);
                pf1 = a.h_prev[e];
                pf2 = a.c_prev[e];
            }
        } else if (MODE == MODE_BWD_CELL && !a.plain) {
            const int j = n0 + e_col;
            if (j < a.H) {
                const int64_t e = (int64_t)e_b * a.H + j;
                pf0 = a.carry[e];
                pf1 = a.c_new[e];
                if (a.pre) pf6 = a.pre[e];
            }
        } else if (n0 + e_col < a.N) {
            // plain product: the accumulated-into output and the bias are requested now as well (the score product of a
            // decode step adds onto scores that were hoisted milliseconds ago: a cold read that used to sit behind the
            // operand loop)
            if (a.beta) pf0 = a.C[(int64_t)e_b * a.ldc + n0 + e_col];
            if (a.bias) pf1 = a.bias[n0 + e_col];
        }
    }

    MMQG_STAMP(1)
    f32x4 acc_p = {0.f, 0.f, 0.f, 0.f}, acc_m = {0.f, 0.f, 0.f, 0.f};
    // The pair loop is unrolled with compile-time pair indices: indexing the kernel-argument array with
    // a run-time value made hipcc fetch the Pair fields with per-lane global loads and a vmcnt(0) wait
    // in front of every operand load (a dependent-load chain per k-chunk).
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        const int pbeg = pi == 0 ? 0 : (pi == 1 ? a.cs1 : a.cs2);
        const int pend = pi == 0 ? min(a.cs1, a.chunks) : (pi == 1 ? min(a.cs2, a.chunks) : a.chunks);
        const int lo = max(q0, pbeg), hi = min(q1, pend);
        if (lo >= hi) continue;
        const float* __restrict__ Ap = a.p[pi].A + (int64_t)ra * a.p[pi].lda;
        const float* __restrict__ Bp = a.p[pi].B + (int64_t)nb * a.p[pi].ldb;
        const int K = a.p[pi].K;
        const bool masked = (MODE == MODE_BWD_CELL) && a.p[pi].masked;
        // k-chunks in flight per wave (2*U 16-byte loads per latency exposure); 4 for the 4-wave forward variant,
        // whose register budget is what lets six workgroups share a CU
        constexpr int U = (MODE == MODE_FWD_CELL && KS == 4) ? 4 : 8;
        for (int q = lo; q < hi; q += U) {
            float4 av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = (q + u - pbeg) * 16 + 4 * ls;
                av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                bv[u] = av[u];
                if (q + u < hi && k < K) {
                    av[u] = *reinterpret_cast<const float4*>(Ap + k);
                    bv[u] = *reinterpret_cast<const float4*>(Bp + k);
                }
            }
            // all lane permutes of the group first, then the MFMAs: left to itself hipcc pairs every
            // MFMA with its two ds_bpermute and an lgkmcnt(0) wait, i.e. one LDS round trip per MFMA
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                av[u].x = __shfl(av[u].x, src_lane, 64); bv[u].x = __shfl(bv[u].x, src_lane, 64);
                av[u].y = __shfl(av[u].y, src_lane, 64); bv[u].y = __shfl(bv[u].y, src_lane, 64);
                av[u].z = __shfl(av[u].z, src_lane, 64); bv[u].z = __shfl(bv[u].z, src_lane, 64);
                av[u].w = __shfl(av[u].w, src_lane, 64); bv[u].w = __shfl(bv[u].w, src_lane, 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (masked) {
                    acc_m = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc_m, 0, 0, 0);
                    acc_m = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc_m, 0, 0, 0);
                    acc_m = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc_m, 0, 0, 0);
                    acc_m = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc_m, 0, 0, 0);
                } else {
                    acc_p = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc_p, 0, 0, 0);
                    acc_p = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc_p, 0, 0, 0);
                    acc_p = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc_p, 0, 0, 0);
                    acc_p = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc_p, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MODE == MODE_BWD_CELL && !a.plain && threadIdx.x < 256 && e_b < a.M && n0 + e_col < a.H) {
        // the remaining epilogue operands are fetched here, behind the partial-tile exchange: held across the
        // operand loop they cost the registers of the fourth wave per SIMD (= two 8-wave workgroups per CU).  In
        // an isolated chain this is slightly slower, inside the two-stream training step it measured +1.2%.
        const int64_t e = (int64_t)e_b * a.H + n0 + e_col;
        pf2 = a.dc[e];
        pf3 = a.c_prev[e];
        if (a.above) pf4 = a.above[(int64_t)e_b * a.above_stride_b + n0 + e_col];
        if (a.extra) pf5 = a.extra[(int64_t)e_b * a.extra_stride_b + n0 + e_col];
    }
    // C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[0][wave][kq * 4 + r][c] = acc_p[r];
        if (MODE == MODE_BWD_CELL) part[1][wave][kq * 4 + r][c] = acc_m[r];
    }
    MMQG_STAMP(2)
    __syncthreads();
    MMQG_STAMP(3)
    const int tid = threadIdx.x;
    if (tid >= 256) { if (!fwd_cell) return; }
    const int row = (tid >> 4) & 15, col = tid & 15;
    float s = 0.f, sm = 0.f;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < KS; ++w) {
            s += part[0][w][row][col];
            if (MODE == MODE_BWD_CELL) sm += part[1][w][row][col];
        }
    }
    const int b = m0 + row;

    if (MODE == MODE_PLAIN || a.plain) {
        const int n = n0 + col;
        if (b < a.M && n < a.N) a.C[(int64_t)b * a.ldc + n] = s + pf0 + pf1;      // (row, col) == (e_row, e_col): prefetched above
        return;
    }

    if (MODE == MODE_FWD_CELL) {
        const int H = a.H;
        const int g = col >> 2, j = n0 + (col & 3);
        const float pre = s + pf0 + pf3 + pf4;
        __syncthreads();                        // all partial sums consumed: reuse part[0][0] for the tile
        if (tid < 256) part[0][0][row][col] = pre;
        __syncthreads();
        if (tid >= 256 || b >= a.M) return;
        const bool active = a.lens ? (a.t < a.lens[b]) : true;
        float* grow = a.gates + (int64_t)b * 4 * H;
        // each of the 4 gate lanes of (b, j) stores its own activated gate; lane g == 0 also does the state
        const float gi = sigmoidf_(part[0][0][row][0 + (col & 3)]);
        const float gf = sigmoidf_(part[0][0][row][4 + (col & 3)]);
        const float gg = tanhf(part[0][0][row][8 + (col & 3)]);
        const float go = sigmoidf_(part[0][0][row][12 + (col & 3)]);
        const float mine = g == 0 ? gi : (g == 1 ? gf : (g == 2 ? gg : go));
        grow[g * H + j] = active ? mine : 0.f;
        if (g != 0) return;
        const int64_t e = (int64_t)b * H + j;
        const float hp = pf1, cp = pf2;
        float h = hp, cc = cp;
        if (active) {
            cc = gf * cp + gi * gg;
            h = go * tanhf(cc);
        }
        a.h_out[e] = h;
        a.c_out[e] = cc;
        if (a.h_drop) a.h_drop[e] = active ? h * dropout_scale(eff_seed(a.seed, a.seed_off), a.stream_id, (uint64_t)e, a.drop_p) : 0.f;
        if (a.y_out) a.y_out[(int64_t)b * a.y_stride_b + j] = active ? h : 0.f;
        MMQG_STAMP(4)
        return;
    }

    // MODE_BWD_CELL
    {
        const int H = a.H;
        const int j = n0 + col;
        if (b >= a.M || j >= H) return;
        const int64_t e = (int64_t)b * H + j;
        float mask = 1.f;
        bool have_mask = false;
        float dh = s + pf0 + pf6;
        if (a.p[1].masked || a.p[2].masked || a.p[0].masked) {
            if (a.drop_p > 0.f) { mask = dropout_scale(eff_seed(a.seed, a.seed_off), a.stream_id, (uint64_t)e, a.drop_p); }
            have_mask = true;
            dh += sm * mask;
        }
        const bool active = a.lens ? (a.t < a.lens[b]) : true;
        float* dg = a.dgates + (int64_t)b * 4 * H;
        if (!active) {
            dg[j] = 0.f; dg[H + j] = 0.f; dg[2 * H + j] = 0.f; dg[3 * H + j] = 0.f;
            a.carry[e] = dh;                    // state was carried forward, so is its gradient
            return;
        }
        if (a.above) {
            if (!have_mask && a.drop_p > 0.f) mask = dropout_scale(eff_seed(a.seed, a.seed_off), a.stream_id, (uint64_t)e, a.drop_p);
            dh += pf4 * mask;
        }
        if (a.extra) dh += pf5;
        const float* gr = a.gates_act + (int64_t)b * 4 * H;
        const float gi = gr[j], gf = gr[H + j], gg = gr[2 * H + j], go = gr[3 * H + j];
        const float tc = tanhf(pf1);
        const float dct = pf2 + dh * go * (1.f - tc * tc);
        dg[j] = dct * gg * gi * (1.f - gi);
        dg[H + j] = dct * pf3 * gf * (1.f - gf);
        dg[2 * H + j] = dct * gi * (1.f - gg * gg);
        dg[3 * H + j] = dh * tc * go * (1.f - go);
        a.dc[e] = dct * gf;
        a.carry[e] = 0.f;
        MMQG_STAMP(4)
    }
}

// Wide forward layer-step for large batches / wide layers (config 5: B = 128, H = 1024).  With one 16 x 16 tile per
// workgroup every workgroup re-reads 16 activation rows and 16 weight rows over the whole K: at M = 128, N = 4096,
// K = 2048 that is 2048 workgroups x 256 KB = 512 MB through the L2s per layer-step (97 us; the arithmetic is 2 GFLOP).
// Here a 4-wave workgroup owns 64 rows x 8 hidden units (32 gate columns): each wave still takes a quarter of K, but
// holds 4 x 2 accumulator tiles, so a weight fragment is used for four row tiles and an activation fragment for two
// column tiles: 256 workgroups x 768 KB = 192 MB.  Same operand path as skinny_kernel (16-byte quad loads straight to
// registers, one ds_bpermute per dword into the MFMA slots, v_mfma_f32_16x16x4_f32), partial tiles through LDS, then
// a thread finishes two (row, unit) pairs: sum of the k-slices + hoisted pre-activations + biases -> cell update.
constexpr int kWideRows = 64, kWideUnits = 8;

// NW waves share a tile's K
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void cell_fwd_wide_kernel(SkinnyBatch batch) {
    __shared__ float part[NW][kWideRows][33];
    const SkinnyK& a = batch.job[blockIdx.z];
    const int m0 = blockIdx.y * kWideRows, n0 = blockIdx.x * kWideUnits;
    if (m0 >= a.M || n0 >= a.H) return;       // jobs of one launch may differ in size
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int lr = lane >> 2, ls = lane & 3;
    const int src_lane = 4 * c + kq;
    const int H = a.H;
    int ra[4], nb[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) ra[r] = min(m0 + 16 * r + lr, a.M - 1);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) nb[cc] = (lr >> 2) * H + min(n0 + 4 * cc + (lr & 3), H - 1);
    const int per = (a.chunks + NW - 1) / NW;
    const int q0 = wave * per, q1 = min(a.chunks, q0 + per);
    f32x4 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) acc[r][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        const int pbeg = pi == 0 ? 0 : (pi == 1 ? a.cs1 : a.cs2);
        const int pend = pi == 0 ? min(a.cs1, a.chunks) : (pi == 1 ? min(a.cs2, a.chunks) : a.chunks);
        const int lo = max(q0, pbeg), hi = min(q1, pend);
        if (lo >= hi) continue;
        const float* __restrict__ Ap[4];
        const float* __restrict__ Bp[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) Ap[r] = a.p[pi].A + (int64_t)ra[r] * a.p[pi].lda;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) Bp[cc] = a.p[pi].B + (int64_t)nb[cc] * a.p[pi].ldb;
        const int K = a.p[pi].K;
        // One k-chunk (32 MFMAs) per step, THREE operand sets taking turns (no register copies; see wide_bwd_kernel).  Config 5,
        // same box, ms per step: two sets of two chunks with copies 21.00, 3 sets 20.64-20.71, 4 sets 20.59-20.80, 6 sets
        // 20.81-20.90, 8 sets 21.55.  Branch-free: clamped addresses, chunks past the range zeroed after the load.
        constexpr int NB = 3;
        float4 fa[NB][4], fb[NB][2];
        auto fetch = [&](float4 (&xa)[4], float4 (&xb)[2], int q) {
            const int k = max(min((min(q, hi - 1) - pbeg) * 16 + 4 * ls, K - 4), 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) xa[r] = *reinterpret_cast<const float4*>(Ap[r] + k);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) xb[cc] = *reinterpret_cast<const float4*>(Bp[cc] + k);
        };
        auto step = [&](float4 (&av)[4], float4 (&bv)[2], int q) {
            if (!((q - pbeg) * 16 + 4 * ls < K)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) av[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) bv[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                av[r].x = __shfl(av[r].x, src_lane, 64); av[r].y = __shfl(av[r].y, src_lane, 64);
                av[r].z = __shfl(av[r].z, src_lane, 64); av[r].w = __shfl(av[r].w, src_lane, 64);
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                bv[cc].x = __shfl(bv[cc].x, src_lane, 64); bv[cc].y = __shfl(bv[cc].y, src_lane, 64);
                bv[cc].z = __shfl(bv[cc].z, src_lane, 64); bv[cc].w = __shfl(bv[cc].w, src_lane, 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    acc[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].x, bv[cc].x, acc[r][cc], 0, 0, 0);
                    acc[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].y, bv[cc].y, acc[r][cc], 0, 0, 0);
                    acc[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].z, bv[cc].z, acc[r][cc], 0, 0, 0);
                    acc[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].w, bv[cc].w, acc[r][cc], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int sx = 0; sx < NB - 1; ++sx) fetch(fa[sx], fb[sx], lo + sx);
        for (int q = lo; q < hi; q += NB) {
#pragma unroll
            for (int sx = 0; sx < NB; ++sx) {
                if (q + sx >= hi) break;                                                 // (uniform)
                fetch(fa[(sx + NB - 1) % NB], fb[(sx + NB - 1) % NB], q + sx + NB - 1);      // the set the previous step freed
                step(fa[sx], fb[sx], q + sx);
            }
        }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[wave][16 * r + 4 * kq + e][16 * cc + c] = acc[r][cc][e];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8 / NW; ++it) {
        const int pp = threadIdx.x + 64 * NW * it;
        const int row = pp >> 3, u = pp & 7;
        const int b = m0 + row, j = n0 + u;
        if (b >= a.M || j >= H) continue;
        const int cb = 16 * (u >> 2) + (u & 3);
        float* grow = a.gates + (int64_t)b * 4 * H;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = part[0][row][cb + 4 * g] + part[1][row][cb + 4 * g] + part[2][row][cb + 4 * g] + part[3][row][cb + 4 * g];
            if (NW == 8) v += part[4 % NW][row][cb + 4 * g] + part[5 % NW][row][cb + 4 * g] + part[6 % NW][row][cb + 4 * g] + part[7 % NW][row][cb + 4 * g];
            if (a.gates_has_pre) v += grow[g * H + j];
            if (a.bias1) v += a.bias1[g * H + j];
            if (a.bias2) v += a.bias2[g * H + j];
            pre[g] = v;
        }
        const bool active = a.lens ? (a.t < a.lens[b]) : true;
        const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
        grow[j] = active ? gi : 0.f;
        grow[H + j] = active ? gf : 0.f;
        grow[2 * H + j] = active ? gg : 0.f;
        grow[3 * H + j] = active ? go : 0.f;
        const int64_t e = (int64_t)b * H + j;
        const float hp = a.h_prev[e], cp = a.c_prev[e];
        float h = hp, cc = cp;
        if (active) {
            cc = gf * cp + gi * gg;
            h = go * tanhf(cc);
        }
        a.h_out[e] = h;
        a.c_out[e] = cc;
        if (a.h_drop) a.h_drop[e] = active ? h * dropout_scale(eff_seed(a.seed, a.seed_off), a.stream_id, (uint64_t)e, a.drop_p) : 0.f;
        if (a.y_out) a.y_out[(int64_t)b * a.y_stride_b + j] = active ? h : 0.f;
    }
}


// Wide BACKWARD layer-step / plain product for batches over 64 rows (config 5: B = 128, H = 1024).  A
// backward layer-step there is dh[128 x 1024] over K = 8192: 16 x 16 tiles make 512 workgroups pull 1 MB each through the
// L2s (512 MB per layer-step, 91 us for 2.1 GFLOP).  Here a 4-wave workgroup owns 64 rows x 32 columns (a weight fragment
// serves four row tiles, an activation fragment two column tiles: 0.375 MB per 16 x 16 output tile instead of 1 MB) and —
// because 64 x 32 tiles alone leave only 64 workgroups per layer-step — the K range of a tile is cut into `ksl` slices
// handled by different workgroups.  The slices meet WITHOUT atomics on the data and without fences: every workgroup
// stores its partial tile(s) write-through (sc1), every wave drains its stores, one lane takes a ticket with a returning
// agent-scope atomic, and the workgroup whose ticket is the last one loads the other slices with sc1 loads, sums them in
// slice order (deterministic) and runs the epilogue (cell backward, or the plain product's bias / accumulate) for the
// tile (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md valid-forms table, first row).  The earlier attempt
// with f32 atomics into a scratch tile plus a ticket lost to the 16 x 16 kernel.
constexpr int kBwRows = 64, kBwCols = 32;

__device__ __forceinline__ float ldg_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void wide_bwd_kernel(SkinnyBatch batch) {
    __shared__ float part[2][4][kBwRows][kBwCols + 1];
    __shared__ unsigned last_flag;
    const SkinnyK& a = batch.job[blockIdx.z];
    const int ksl = batch.ksl;
    const int tn = blockIdx.x / ksl, slice = blockIdx.x - tn * ksl;
    const int m0 = blockIdx.y * kBwRows, n0 = tn * kBwCols;
    if (m0 >= a.M || n0 >= a.N) return;       // jobs of one launch may differ in size
    const bool cell = MODE == MODE_BWD_CELL && !a.plain;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int lr = lane >> 2, ls = lane & 3;
    const int src_lane = 4 * c + kq;
    int ra[4], nb[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) ra[r] = min(m0 + 16 * r + lr, a.M - 1);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) nb[cc] = min(n0 + 16 * cc + lr, a.N - 1);
    // this workgroup's k-chunks, then this wave's quarter of them
    const int per_s = (a.chunks + ksl - 1) / ksl;
    const int s0 = slice * per_s, s1 = min(a.chunks, s0 + per_s);
    const int per = (max(s1 - s0, 0) + 3) / 4;
    const int q0 = s0 + wave * per, q1 = min(s1, q0 + per);
    f32x4 accp[4][2], accm[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { accp[r][cc] = f32x4{0.f, 0.f, 0.f, 0.f}; accm[r][cc] = accp[r][cc]; }
    bool any_masked = false;
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        const int pbeg = pi == 0 ? 0 : (pi == 1 ? a.cs1 : a.cs2);
        const int pend = pi == 0 ? min(a.cs1, a.chunks) : (pi == 1 ? min(a.cs2, a.chunks) : a.chunks);
        const bool masked = cell && a.p[pi].masked;
        any_masked = any_masked || (masked && pend > pbeg);
        const int lo = max(q0, pbeg), hi = min(q1, pend);
        if (lo >= hi) continue;
        const float* __restrict__ Ap[4];
        const float* __restrict__ Bp[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) Ap[r] = a.p[pi].A + (int64_t)ra[r] * a.p[pi].lda;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) Bp[cc] = a.p[pi].B + (int64_t)nb[cc] * a.p[pi].ldb;
        const int K = a.p[pi].K;
        // One k-chunk (32 MFMAs) per step, THREE operand sets taking turns (no register copies): the 6 sixteen-byte loads of
        // each of the next two steps are in flight behind a step's MFMAs.  Before: two sets of two chunks, 64 MFMAs per
        // step and 24 v_mov per step to hand the next set over.  Deeper rotations lost again (config 5, same box, ms per
        // step: 3 sets 20.51-20.53, 4 sets 20.52-20.54, 5 sets 20.59-20.61): what pays is the finer step without the
        // copies, not the depth.  No branch around a load: addresses are clamped and out-of-range chunks are zeroed after
        // the load.
        constexpr int NB = 3;
        float4 fa[NB][4], fb[NB][2];
        auto fetch = [&](float4 (&xa)[4], float4 (&xb)[2], int q) {
            const int k = min((min(q, hi - 1) - pbeg) * 16 + 4 * ls, K - 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) xa[r] = *reinterpret_cast<const float4*>(Ap[r] + max(k, 0));
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) xb[cc] = *reinterpret_cast<const float4*>(Bp[cc] + max(k, 0));
        };
        auto step = [&](float4 (&av)[4], float4 (&bv)[2], int q) {
            if (!((q - pbeg) * 16 + 4 * ls < K)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) av[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) bv[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                av[r].x = __shfl(av[r].x, src_lane, 64); av[r].y = __shfl(av[r].y, src_lane, 64);
                av[r].z = __shfl(av[r].z, src_lane, 64); av[r].w = __shfl(av[r].w, src_lane, 64);
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                bv[cc].x = __shfl(bv[cc].x, src_lane, 64); bv[cc].y = __shfl(bv[cc].y, src_lane, 64);
                bv[cc].z = __shfl(bv[cc].z, src_lane, 64); bv[cc].w = __shfl(bv[cc].w, src_lane, 64);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (masked) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        accm[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].x, bv[cc].x, accm[r][cc], 0, 0, 0);
                        accm[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].y, bv[cc].y, accm[r][cc], 0, 0, 0);
                        accm[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].z, bv[cc].z, accm[r][cc], 0, 0, 0);
                        accm[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].w, bv[cc].w, accm[r][cc], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        accp[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].x, bv[cc].x, accp[r][cc], 0, 0, 0);
                        accp[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].y, bv[cc].y, accp[r][cc], 0, 0, 0);
                        accp[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].z, bv[cc].z, accp[r][cc], 0, 0, 0);
                        accp[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].w, bv[cc].w, accp[r][cc], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int sx = 0; sx < NB - 1; ++sx) fetch(fa[sx], fb[sx], lo + sx);
        for (int q = lo; q < hi; q += NB) {
#pragma unroll
            for (int sx = 0; sx < NB; ++sx) {
                if (q + sx >= hi) break;                                                 // (uniform)
                fetch(fa[(sx + NB - 1) % NB], fb[(sx + NB - 1) % NB], q + sx + NB - 1);      // the set the previous step freed
                step(fa[sx], fb[sx], q + sx);
            }
        }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                part[0][wave][16 * r + 4 * kq + e][16 * cc + c] = accp[r][cc][e];
                part[1][wave][16 * r + 4 * kq + e][16 * cc + c] = accm[r][cc][e];
            }
    __syncthreads();
    // thread -> (row, 8 consecutive columns) of the 64 x 32 tile
    const int row = threadIdx.x >> 2, c8 = (threadIdx.x & 3) * 8;
    f32x4 sp[2], sm[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = c8 + 4 * h + i;
            sp[h][i] = part[0][0][row][cc] + part[0][1][row][cc] + part[0][2][row][cc] + part[0][3][row][cc];
            sm[h][i] = part[1][0][row][cc] + part[1][1][row][cc] + part[1][2][row][cc] + part[1][3][row][cc];
        }
    if (ksl > 1) {
        // ---- the slices of this tile meet: write-through partials (16-byte stores), drained, one ticket; the last one
        // requests every slice's tile at once and sums in slice order
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int64_t tile_id = ((int64_t)blockIdx.z * batch.ws_tiles_m + blockIdx.y) * batch.ws_tiles_n + tn;
        constexpr int kRec = 2 * kBwRows * kBwCols;                       // floats per (tile, slice): plain | masked
        const int tile_bytes = ksl * kRec * 4;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(batch.ws + tile_id * ksl * kRec, 0, tile_bytes, 0x00020000);
        const int my = (slice * kRec + row * kBwCols + c8) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sp[h]), rs, my + 16 * h, 0, 16);
            if (any_masked) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sm[h]), rs, my + kBwRows * kBwCols * 4 + 16 * h, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(batch.tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = old == (unsigned)ksl - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (!last_flag) return;
        f32x4 vp[4][2], vm[4][2];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int o = (min(sl, ksl - 1) * kRec + row * kBwCols + c8) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) vp[sl][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o + 16 * h, 0, 16));
        }
        if (any_masked) {                      // (uniform)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int o = (min(sl, ksl - 1) * kRec + row * kBwCols + c8) * 4 + kBwRows * kBwCols * 4;
#pragma unroll
                for (int h = 0; h < 2; ++h) vm[sl][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o + 16 * h, 0, 16));
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) { vm[sl][0] = f32x4{0.f, 0.f, 0.f, 0.f}; vm[sl][1] = vm[sl][0]; }
        }
        // (every slice's tile in flight before the first is added: left alone the scheduler pairs each of the up to 16 loads —
        // cross-XCD, write-through data: a fabric round trip each — with its add, load, s_waitcnt vmcnt(0), add)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) { sp[h] = f32x4{0.f, 0.f, 0.f, 0.f}; sm[h] = sp[h]; }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const float w = sl < ksl ? 1.f : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) { sp[h] += w * vp[sl][h]; sm[h] += w * vm[sl][h]; }
        }
        if (threadIdx.x == 0) __hip_atomic_store(batch.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int b = m0 + row;
    if (b >= a.M) return;
    if (!cell) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + c8 + 4 * h + i;
                if (n < a.N) {
                    float* dst = a.C + (int64_t)b * a.ldc + n;
                    float v = sp[h][i] + (a.bias ? a.bias[n] : 0.f);
                    if (a.beta) v += *dst;
                    *dst = v;
                }
            }
        return;
    }
    {
        const int H = a.H;
        const bool active = a.lens ? (a.t < a.lens[b]) : true;
        float* dg = a.dgates + (int64_t)b * 4 * H;
        const float* gr = a.gates_act + (int64_t)b * 4 * H;
        const uint64_t seed = a.drop_p > 0.f ? eff_seed(a.seed, a.seed_off) : 0;
        const bool need_mask = (any_masked || a.above) && a.drop_p > 0.f;
        // whole 4-column groups with 16-byte aligned operands (every shape the executors produce): all operands of the
        // thread's 8 elements are requested before the first is used
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
        const bool vec = (H % 4 == 0) && n0 + c8 + 8 <= H && !a.above && al16(a.carry) && al16(a.c_new) && al16(a.c_prev) &&
                         al16(a.dc) && al16(a.gates_act) && al16(a.dgates) && (!a.pre || al16(a.pre)) &&
                         (!a.extra || (al16(a.extra) && a.extra_stride_b % 4 == 0));
        if (vec) {
            f32x4 car[2], pre[2], ext[2], cn[2], cp[2], dcv[2], g4[4][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = n0 + c8 + 4 * h;
                const int64_t e = (int64_t)b * H + j;
                car[h] = *reinterpret_cast<const f32x4*>(a.carry + e);
                pre[h] = a.pre ? *reinterpret_cast<const f32x4*>(a.pre + e) : f32x4{0.f, 0.f, 0.f, 0.f};
                ext[h] = (a.extra && active) ? *reinterpret_cast<const f32x4*>(a.extra + (int64_t)b * a.extra_stride_b + j) : f32x4{0.f, 0.f, 0.f, 0.f};
                cn[h] = *reinterpret_cast<const f32x4*>(a.c_new + e);
                cp[h] = *reinterpret_cast<const f32x4*>(a.c_prev + e);
                dcv[h] = *reinterpret_cast<const f32x4*>(a.dc + e);
#pragma unroll
                for (int g = 0; g < 4; ++g) g4[g][h] = *reinterpret_cast<const f32x4*>(gr + g * H + j);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = n0 + c8 + 4 * h;
                const int64_t e = (int64_t)b * H + j;
                f32x4 mask = f32x4{1.f, 1.f, 1.f, 1.f};
                if (need_mask) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) mask[i] = dropout_scale(seed, a.stream_id, (uint64_t)(e + i), a.drop_p);
                }
                f32x4 dh = sp[h] + car[h] + pre[h];
                if (any_masked) dh += sm[h] * mask;
                if (!active) {
                    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dg + g * H + j) = z;
                    *reinterpret_cast<f32x4*>(a.carry + e) = dh;         // state was carried forward, so is its gradient
                    continue;
                }
                dh += ext[h];
                const f32x4 gi = g4[0][h], gf = g4[1][h], gg = g4[2][h], go = g4[3][h];
                f32x4 tc;
#pragma unroll
                for (int i = 0; i < 4; ++i) tc[i] = tanhf(cn[h][i]);
                const f32x4 dct = dcv[h] + dh * go * (1.f - tc * tc);
                *reinterpret_cast<f32x4*>(dg + j) = dct * gg * gi * (1.f - gi);
                *reinterpret_cast<f32x4*>(dg + H + j) = dct * cp[h] * gf * (1.f - gf);
                *reinterpret_cast<f32x4*>(dg + 2 * H + j) = dct * gi * (1.f - gg * gg);
                *reinterpret_cast<f32x4*>(dg + 3 * H + j) = dh * tc * go * (1.f - go);
                *reinterpret_cast<f32x4*>(a.dc + e) = dct * gf;
                *reinterpret_cast<f32x4*>(a.carry + e) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = n0 + c8 + 4 * h + i;
                if (j >= H) continue;
                const int64_t e = (int64_t)b * H + j;
                float mask = 1.f;
                if (need_mask) mask = dropout_scale(seed, a.stream_id, (uint64_t)e, a.drop_p);
                float dh = sp[h][i] + a.carry[e] + (a.pre ? a.pre[e] : 0.f);
                if (any_masked) dh += sm[h][i] * mask;
                if (!active) {
                    dg[j] = 0.f; dg[H + j] = 0.f; dg[2 * H + j] = 0.f; dg[3 * H + j] = 0.f;
                    a.carry[e] = dh;                    // state was carried forward, so is its gradient
                    continue;
                }
                if (a.above) dh += a.above[(int64_t)b * a.above_stride_b + j] * mask;
                if (a.extra) dh += a.extra[(int64_t)b * a.extra_stride_b + j];
                const float gi = gr[j], gf = gr[H + j], gg = gr[2 * H + j], go = gr[3 * H + j];
                const float tc = tanhf(a.c_new[e]);
                const float dct = a.dc[e] + dh * go * (1.f - tc * tc);
                dg[j] = dct * gg * gi * (1.f - gi);
                dg[H + j] = dct * a.c_prev[e] * gf * (1.f - gf);
                dg[2 * H + j] = dct * gi * (1.f - gg * gg);
                dg[3 * H + j] = dh * tc * go * (1.f - go);
                a.dc[e] = dct * gf;
                a.carry[e] = 0.f;
            }
    }
}

// dst[c][r] = src[r][c]   (rows x cols -> cols x rows), 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int ld_src, int rows, int cols,
                                                        float* __restrict__ dst, int ld_dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, cc = c0 + tx;
        tile[ty + i][tx] = (r < rows && cc < cols) ? src[(int64_t)r * ld_src + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int cc = c0 + ty + i, r = r0 + tx;
        if (cc < cols && r < rows) dst[(int64_t)cc * ld_dst + r] = tile[tx][ty + i];
    }
}

constexpr int kMaxTranspose = 16;
struct TransposeBatch { mmqg_transpose_job job[kMaxTranspose]; };

// 64 x 64 tiles, 16-byte accesses on both sides (256 contiguous bytes per row segment): for jobs whose pointers and
// leading dimensions are 16-byte aligned and whose column count is a multiple of 4 (every weight matrix of the model); the
// 32 x 32 scalar kernel below had 128-byte row segments and reached 2.1 TB/s
__global__ __launch_bounds__(256) void transpose_batch64_kernel(TransposeBatch b) {
    __shared__ float tile[64][65];
    const mmqg_transpose_job& j = b.job[blockIdx.z];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    if (r0 >= j.rows || c0 >= j.cols) return;      // the grid is the bounding box of the jobs
    const int q = threadIdx.x & 15, p = threadIdx.x >> 4;      // 16 float4 lanes x 16 rows per pass
#pragma unroll
    for (int i = 0; i < 64; i += 16) {
        const int r = r0 + p + i, cc = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < j.rows && cc < j.cols) v = *reinterpret_cast<const float4*>(j.src + (int64_t)r * j.ld_src + cc);
        tile[p + i][4 * q] = v.x; tile[p + i][4 * q + 1] = v.y; tile[p + i][4 * q + 2] = v.z; tile[p + i][4 * q + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 64; i += 16) {
        const int cc = c0 + p + i, r = r0 + 4 * q;
        if (cc < j.cols && r + 3 < j.rows) {
            *reinterpret_cast<float4*>(j.dst + (int64_t)cc * j.ld_dst + r) =
                make_float4(tile[4 * q][p + i], tile[4 * q + 1][p + i], tile[4 * q + 2][p + i], tile[4 * q + 3][p + i]);
        } else if (cc < j.cols) {               // the last rows of a matrix whose row count is not a multiple of 4
            for (int k = 0; k < 4 && r + k < j.rows; ++k) j.dst[(int64_t)cc * j.ld_dst + r + k] = tile[4 * q + k][p + i];
        }
    }
}

__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch b) {
    __shared__ float tile[32][33];
    const mmqg_transpose_job& j = b.job[blockIdx.z];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    if (r0 >= j.rows || c0 >= j.cols) return;      // the grid is the bounding box of the jobs
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, cc = c0 + tx;
        tile[ty + i][tx] = (r < j.rows && cc < j.cols) ? j.src[(int64_t)r * j.ld_src + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int cc = c0 + ty + i, r = r0 + tx;
        if (cc < j.cols && r < j.rows) j.dst[(int64_t)cc * j.ld_dst + r] = tile[tx][ty + i];
    }
}

bool pair_ok(const mmqg::SkinnyPair& p) {
    return p.A && p.B && p.K > 0 && p.K % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 && mmqg::aligned16(p.A) &&
           mmqg::aligned16(p.B);
}

// workspace of the wide backward / plain kernel, set for the duration of an executor call (sequence.hip)
thread_local float* t_wide_ws = nullptr;
thread_local int64_t t_wide_bytes = 0;

// The tickets sit at a FIXED place — the first kWideTicketBytes of the workspace — whatever the launch's tile grid
// and slice count: launches of different geometry share one workspace, and a ticket word that another launch's
// partial tiles had overwritten would never count up to ksl.
constexpr int kWideTicketBytes = 8192;
inline int64_t wide_ws_need(int njobs, int tiles_m, int tiles_n, int ksl) {
    if ((int64_t)njobs * tiles_m * tiles_n * 4 > kWideTicketBytes) return -1;
    return kWideTicketBytes + (int64_t)njobs * tiles_m * tiles_n * ksl * 2 * kBwRows * kBwCols * 4;
}

// batches over 64 rows: 64 x 32 tiles with k slices (wide_bwd_kernel); 0 = launched, 1 = not taken
template <int MODE>
int try_wide_bwd(const SkinnyBatch& b0, int njobs, hipStream_t s, const char* what) {
    static const bool off = [] { const char* e = getenv("MMQG_NO_WIDE_BWD"); return e && atoi(e) != 0; }();
    if (off || !t_wide_ws) return 1;
    int maxM = 0, maxN = 0, min_chunks = 1 << 30;
    int64_t tiles = 0;
    for (int i = 0; i < njobs; ++i) {
        const SkinnyK& k = b0.job[i];
        if (k.M <= 64) return 1;                         // the 16 x 16 kernel serves small batches
        maxM = std::max(maxM, k.M); maxN = std::max(maxN, k.N);
        min_chunks = std::min(min_chunks, k.chunks);
        tiles += (int64_t)mmqg::ceil_div(k.M, kBwRows) * mmqg::ceil_div(k.N, kBwCols);
    }
    const int tiles_m = mmqg::ceil_div(maxM, kBwRows), tiles_n = mmqg::ceil_div(maxN, kBwCols);
    // k slices: enough workgroups for two per CU, at least 32 k-chunks (512 k) per slice
    static const int max_ksl = [] { const char* e = getenv("MMQG_WIDE_MAX_KSL"); return e ? atoi(e) : 4; }();
    static const int target_wgs = [] { const char* e = getenv("MMQG_WIDE_TARGET_WGS"); return e ? atoi(e) : 256; }();
    int ksl = 1;
    while (ksl < max_ksl && tiles * ksl < target_wgs && min_chunks / (2 * ksl) >= 32) ksl *= 2;
    const int64_t need = wide_ws_need(njobs, tiles_m, tiles_n, ksl);
    if (need < 0 || need > t_wide_bytes || !mmqg::aligned16(t_wide_ws)) return 1;
    SkinnyBatch b = b0;
    b.ksl = ksl; b.ws = t_wide_ws + kWideTicketBytes / 4; b.tickets = reinterpret_cast<unsigned*>(t_wide_ws);
    b.ws_tiles_n = tiles_n; b.ws_tiles_m = tiles_m;
    hipLaunchKernelGGL((wide_bwd_kernel<MODE>), dim3(tiles_n * ksl, tiles_m, njobs), dim3(256), 0, s, b);
    return mmqg::check_launch(what);
}

template <int MODE>
int launch_skinny_batch(const SkinnyBatch& b, int njobs, hipStream_t s, const char* what) {
    if (MODE != MODE_FWD_CELL) {
        const int rc = try_wide_bwd<MODE == MODE_FWD_CELL ? MODE_PLAIN : MODE>(b, njobs, s, what);
        if (rc <= 0) return rc;
    }
    int tiles_n = 0, tiles_m = 0, chunks = 0;
    int64_t wgs = 0;       // workgroups that do work (the grid is the bounding box of the jobs)
    for (int i = 0; i < njobs; ++i) {
        const SkinnyK& k = b.job[i];
        const int tn = (MODE == MODE_FWD_CELL && !k.plain) ? k.H / 4 : mmqg::ceil_div(k.N, 16), tm = mmqg::ceil_div(k.M, 16);
        tiles_n = std::max(tiles_n, tn);
        tiles_m = std::max(tiles_m, tm);
        wgs += (int64_t)tn * tm;
        chunks = std::max(chunks, k.chunks);
    }
    if (MODE == MODE_FWD_CELL) {
        // large batch / wide layer: the 64-row x 8-unit kernel when its grid still fills the chip
        static const bool no_wide = [] { const char* e = getenv("MMQG_NO_WIDE"); return e && atoi(e) != 0; }();
        int64_t wide_wgs = 0;
        int wx = 0, wy = 0;
        bool ok = !no_wide;
        for (int i = 0; i < njobs && ok; ++i) {
            const SkinnyK& k = b.job[i];
            ok = k.M > 64 && k.H % 4 == 0 && !k.plain;
            const int tx = mmqg::ceil_div(k.H, kWideUnits), ty = mmqg::ceil_div(k.M, kWideRows);
            wide_wgs += (int64_t)tx * ty;
            wx = std::max(wx, tx); wy = std::max(wy, ty);
        }
        if (ok && wide_wgs >= 192) {
            // (eight waves per tile for single-job launches — one workgroup per CU — measured slower: config 5 21.62-21.68
            // against 21.28-21.38 ms)
            hipLaunchKernelGGL(cell_fwd_wide_kernel<4>, dim3(wx, wy, njobs), dim3(256), 0, s, b);
            return mmqg::check_launch(what);
        }
    }
    // (the backward layer-step stays on 16 x 16 tiles at every size: 64 x 32 tiles measured slower both alone — 64
    // workgroups, per-CU fetch rate sets the time: config 5 26.3 vs 23.9 ms — and with k slices across workgroups meeting
    // through f32 atomics and a ticket counter: 31.7 / 38.4 / 46.5 ms at 3 / 5 / 8 slices; 32 x 16 tiles in 8-wave
    // workgroups, two per CU: 24.6 vs 24.2 ms)
    dim3 grid(tiles_n, tiles_m, njobs);
    // 8 k-slices per tile once a tile has >= 64 k-chunks, unless that would put more than ~4096 waves
    // in flight (three layer-steps in one launch): beyond that the extra waves only add fixed cost
    static const int ks8_from = [] { const char* e = getenv("MMQG_SKINNY_KS8_FROM"); return e ? atoi(e) : 64; }();
    if (chunks >= ks8_from && wgs * 8 <= 4096) hipLaunchKernelGGL((skinny_kernel<MODE, 8>), grid, dim3(512), 0, s, b);
    else hipLaunchKernelGGL((skinny_kernel<MODE, 4>), grid, dim3(256), 0, s, b);
    return mmqg::check_launch(what);
}

template <int MODE>
int launch_skinny(const SkinnyK& k, int tiles_n, hipStream_t s, const char* what) {
    (void)tiles_n;
    SkinnyBatch b{};
    b.job[0] = k;
    return launch_skinny_batch<MODE>(b, 1, s, what);
}

int fill_pairs(SkinnyK& k, const mmqg::SkinnyPair* pairs, int npairs, const char* who) {
    MMQG_REQUIRE(npairs >= 1 && npairs <= 3, "%s: need 1..3 operand pairs", who);
    int chunks = 0;
    k.cs1 = k.cs2 = 1 << 30;
    for (int i = 0; i < 3; ++i) {
        if (i < npairs) {
            MMQG_REQUIRE(pair_ok(pairs[i]), "%s: operand pair %d is not 16-byte aligned / K,ld not multiples of 4", who, i);
            k.p[i] = Pair{pairs[i].A, pairs[i].lda, pairs[i].B, pairs[i].ldb, pairs[i].K, pairs[i].masked};
            if (i == 1) k.cs1 = chunks;
            if (i == 2) k.cs2 = chunks;
            chunks += mmqg::ceil_div(pairs[i].K, 16);
        } else {
            k.p[i] = Pair{nullptr, 0, nullptr, 0, 0, 0};
        }
    }
    k.chunks = chunks;
    return 0;
}

}  // namespace

namespace mmqg {

// the caller-owned workspace of the wide backward / plain kernel for the executor call in progress on this thread
// (null = none: batches over 64 rows then take the 16 x 16 kernel); the ticket words at its start must be zero
void skinny_set_wide_ws(float* ws, int64_t bytes) { t_wide_ws = ws; t_wide_bytes = ws ? bytes : 0; }
int64_t skinny_wide_ws_bytes(int M, int max_N) {
    if (M <= 64 || max_N <= 0) return 0;
    const int64_t n = wide_ws_need(3, ceil_div(M, kBwRows), ceil_div(max_N, kBwCols), 4);
    return n < 0 ? 0 : n;
}

bool skinny_usable(const SkinnyPair* pairs, int npairs) {
    for (int i = 0; i < npairs; ++i)
        if (!pair_ok(pairs[i])) return false;
    return npairs >= 1 && npairs <= 3;
}

int skinny_plain(int M, int N, const SkinnyPair* pairs, int npairs, const float* bias, int beta, float* C, int ldc,
                 hipStream_t s) {
    MMQG_REQUIRE(M >= 0 && N >= 0 && C && ldc >= N, "skinny_plain: bad arguments");
    if (M == 0 || N == 0) return 0;
    SkinnyK k{};
    MMQG_TRY(fill_pairs(k, pairs, npairs, "skinny_plain"));
    k.M = M; k.N = N; k.C = C; k.ldc = ldc; k.beta = beta ? 1 : 0; k.bias = bias;
    return launch_skinny<MODE_PLAIN>(k, ceil_div(N, 16), s, "skinny_plain");
}

static int fill_plain_job(SkinnyK& k, const SkinnyPlainJob& j) {
    MMQG_REQUIRE(j.M > 0 && j.N > 0 && j.C && j.ldc >= j.N, "skinny_plain: bad arguments");
    MMQG_TRY(fill_pairs(k, j.pairs, j.npairs, "skinny_plain"));
    k.M = j.M; k.N = j.N; k.C = j.C; k.ldc = j.ldc; k.beta = j.beta ? 1 : 0; k.bias = j.bias; k.plain = 1;
    return 0;
}

int skinny_plain_multi(const SkinnyPlainJob* jobs, int njobs, hipStream_t s) {
    MMQG_REQUIRE(njobs >= 1 && njobs <= 3, "skinny_plain_multi: 1..3 jobs");
    SkinnyBatch b{};
    for (int i = 0; i < njobs; ++i) MMQG_TRY(fill_plain_job(b.job[i], jobs[i]));
    return launch_skinny_batch<MODE_PLAIN>(b, njobs, s, "skinny_plain");
}

static int fill_fwd_job(SkinnyK& k, const SkinnyFwdJob& j) {
    const CellFwd& f = j.cell;
    MMQG_REQUIRE(f.B > 0 && f.H > 0 && f.H % 4 == 0 && f.ld_g == 4 * f.H, "skinny_cell_fwd: need H %% 4 == 0 and compact gates");
    MMQG_REQUIRE(f.gates && f.h_prev && f.c_prev && f.h_out && f.c_out, "skinny_cell_fwd: null pointer");
    MMQG_TRY(fill_pairs(k, j.pairs, j.npairs, "skinny_cell_fwd"));
    k.M = f.B; k.N = 4 * f.H; k.H = f.H;
    k.lens = f.lens; k.t = f.t; k.drop_p = f.p; k.seed = f.seed; k.stream_id = f.stream_id; k.seed_off = f.seed_off;
    k.gates = f.gates; k.gates_has_pre = j.gates_has_pre; k.bias1 = j.bias1; k.bias2 = j.bias2;
    k.h_prev = f.h_prev; k.c_prev = f.c_prev; k.h_out = f.h_out; k.c_out = f.c_out; k.h_drop = f.h_drop;
    k.y_out = f.y_out; k.y_stride_b = f.y_stride_b;
    return 0;
}

static int fill_bwd_job(SkinnyK& k, const SkinnyBwdJob& j) {
    const CellBwd& f = j.cell;
    MMQG_REQUIRE(f.B > 0 && f.H > 0 && f.ld_dg == 4 * f.H, "skinny_cell_bwd: need compact dgates");
    MMQG_REQUIRE(f.gates_act && f.c_prev && f.c_new && f.dh_rec && f.dc && f.dgates, "skinny_cell_bwd: null pointer");
    MMQG_TRY(fill_pairs(k, j.pairs, j.npairs, "skinny_cell_bwd"));
    k.M = f.B; k.N = f.H; k.H = f.H;
    k.lens = f.lens; k.t = f.t; k.drop_p = f.p; k.seed = f.seed; k.stream_id = f.stream_id; k.seed_off = f.seed_off;
    k.gates_act = f.gates_act; k.c_prev = f.c_prev; k.c_new = f.c_new; k.carry = f.dh_rec;
    k.above = f.dh_above; k.above_stride_b = f.above_stride_b; k.extra = f.dh_extra; k.extra_stride_b = f.extra_stride_b;
    k.pre = f.dh_pre;
    k.dc = f.dc; k.dgates = f.dgates;
    return 0;
}

int skinny_cell_fwd_multi(const SkinnyFwdJob* jobs, int njobs, hipStream_t s) {
    MMQG_REQUIRE(njobs >= 1 && njobs <= 3, "skinny_cell_fwd_multi: 1..3 jobs");
    SkinnyBatch b{};
    for (int i = 0; i < njobs; ++i) MMQG_TRY(fill_fwd_job(b.job[i], jobs[i]));
    return launch_skinny_batch<MODE_FWD_CELL>(b, njobs, s, "skinny_cell_fwd");
}

int skinny_cell_bwd_multi(const SkinnyBwdJob* jobs, int njobs, hipStream_t s) {
    MMQG_REQUIRE(njobs >= 1 && njobs <= 3, "skinny_cell_bwd_multi: 1..3 jobs");
    SkinnyBatch b{};
    for (int i = 0; i < njobs; ++i) MMQG_TRY(fill_bwd_job(b.job[i], jobs[i]));
    return launch_skinny_batch<MODE_BWD_CELL>(b, njobs, s, "skinny_cell_bwd");
}

int skinny_cell_bwd_plus(const SkinnyBwdJob& cell, const SkinnyPlainJob* extra, int nextra, hipStream_t s) {
    MMQG_REQUIRE(nextra >= 0 && nextra <= 2, "skinny_cell_bwd_plus: at most 2 extra products");
    SkinnyBatch b{};
    MMQG_TRY(fill_bwd_job(b.job[0], cell));
    for (int i = 0; i < nextra; ++i) MMQG_TRY(fill_plain_job(b.job[1 + i], extra[i]));
    return launch_skinny_batch<MODE_BWD_CELL>(b, 1 + nextra, s, "skinny_cell_bwd");
}

int skinny_cell_fwd_plus(const SkinnyFwdJob& cell, const SkinnyPlainJob* extra, int nextra, hipStream_t s) {
    MMQG_REQUIRE(nextra >= 0 && nextra <= 2, "skinny_cell_fwd_plus: at most 2 extra products");
    SkinnyBatch b{};
    MMQG_TRY(fill_fwd_job(b.job[0], cell));
    for (int i = 0; i < nextra; ++i) MMQG_TRY(fill_plain_job(b.job[1 + i], extra[i]));
    return launch_skinny_batch<MODE_FWD_CELL>(b, 1 + nextra, s, "skinny_cell_fwd");
}

int skinny_cell_fwd(const SkinnyPair* pairs, int npairs, int gates_has_pre, const float* bias1, const float* bias2,
                    const CellFwd& f, hipStream_t s) {
    if (f.B == 0) return 0;
    MMQG_REQUIRE(npairs >= 1 && npairs <= 3, "skinny_cell_fwd: need 1..3 operand pairs");
    SkinnyFwdJob j{};
    for (int i = 0; i < npairs; ++i) j.pairs[i] = pairs[i];
    j.npairs = npairs; j.gates_has_pre = gates_has_pre; j.bias1 = bias1; j.bias2 = bias2; j.cell = f;
    return skinny_cell_fwd_multi(&j, 1, s);
}

int skinny_cell_bwd(const SkinnyPair* pairs, int npairs, const CellBwd& f, hipStream_t s) {
    if (f.B == 0) return 0;
    MMQG_REQUIRE(npairs >= 1 && npairs <= 3, "skinny_cell_bwd: need 1..3 operand pairs");
    SkinnyBwdJob j{};
    for (int i = 0; i < npairs; ++i) j.pairs[i] = pairs[i];
    j.npairs = npairs; j.cell = f;
    return skinny_cell_bwd_multi(&j, 1, s);
}

int transpose_f32_batch(const mmqg_transpose_job* jobs, int n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && (n == 0 || jobs), "transpose_f32_batch: bad arguments");
    for (int i0 = 0; i0 < n; i0 += kMaxTranspose) {
        TransposeBatch b{};
        const int m = std::min(kMaxTranspose, n - i0);
        int gx = 0, gy = 0;
        bool vec = true;
        for (int i = 0; i < m; ++i) {
            const mmqg_transpose_job& j = jobs[i0 + i];
            vec = vec && (j.cols % 4 == 0) && (j.ld_src % 4 == 0) && (j.ld_dst % 4 == 0) && aligned16(j.src) && aligned16(j.dst);
            MMQG_REQUIRE(j.rows >= 0 && j.cols >= 0 && (j.rows == 0 || j.cols == 0 || (j.src && j.dst && j.ld_src >= j.cols && j.ld_dst >= j.rows)),
                         "transpose_f32_batch: bad job %d", i0 + i);
            b.job[i] = j;
            gx = std::max(gx, ceil_div(j.cols, 32)); gy = std::max(gy, ceil_div(j.rows, 32));
        }
        if (gx == 0 || gy == 0) continue;
        if (vec) hipLaunchKernelGGL(transpose_batch64_kernel, dim3((gx + 1) / 2, (gy + 1) / 2, m), dim3(256), 0, s, b);
        else hipLaunchKernelGGL(transpose_batch_kernel, dim3(gx, gy, m), dim3(256), 0, s, b);
        MMQG_TRY(check_launch("transpose_f32_batch"));
    }
    return 0;
}

int transpose_f32(const float* src, int ld_src, int rows, int cols, float* dst, int ld_dst, hipStream_t s) {
    MMQG_REQUIRE(rows >= 0 && cols >= 0 && src && dst && ld_src >= cols && ld_dst >= rows, "transpose_f32: bad arguments");
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(256), 0, s, src, ld_src, rows,
                       cols, dst, ld_dst);
    return check_launch("transpose_f32");
}

}  // namespace mmqg
