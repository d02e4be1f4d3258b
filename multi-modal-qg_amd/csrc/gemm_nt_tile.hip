// C[M,N] = A[M,K] * B[N,K]^T + bias (+ bias2)   — both operands k-contiguous ("NT": activations x Linear weight),
// for LARGE outputs with SHORT K: the vocabulary projection logits = h_top * W_out^T + b (model/decoder.py:106;
// 1280 x 10000 x 512 at config 2, 2560 x 50000 x 1024 at config 5).
//
// Why a second GEMM kernel: on that shape the 128x128-tile kernel (gemm_f32.hip) sits at 0.53 of the fp32 MFMA
// peak although its tile body reaches 0.84 at 4096^3 — tools/bench_gemm_k.py shows the time is 18 us + 0.27 us per
// unit of K, i.e. the GRID is the limit (790 tiles = 3.09 per CU, some CUs run 4; launch + first-tile latency is
// 11% at K = 512).  Here the grid is made to fit the chip instead: 256 x BN tiles with BN (a multiple of 16, <= 208)
// chosen so that the tile count is just under a multiple of the CU count — 245 tiles of 256 x 208 at config 2, ONE
// per CU — and one 8-wave workgroup per CU runs them in a persistent loop.  Measured: 157 -> 128 us at config 2
// (0.53 -> 0.65 of the fp32 MFMA peak; what is left is the 51 MB of logits that every CU writes at the same moment
// — about 10 us, the 128x128 kernel's three rounds hid their stores behind other tiles' arithmetic — plus launch and
// first-chunk latency); at config 5 it ties the tiled kernel (0.78: the loop's asymptote, the chip holds ~2.1 GHz).
// 4-wave workgroups, two per CU (MMQG_NT_WAVES=4), measured 8% slower.
//
// Data path: operands go global -> LDS with the direct-to-LDS load (global_load_lds_dwordx4: no staging registers,
// no ds_write) already in MFMA-fragment order — one wave-instruction moves the 1 KB fragment (16 rows x 16 k) whose
// lane l = (i = l & 15, kq = l >> 4) holds row i's four consecutive k = 16c + 4kq + {0..3}; the four
// v_mfma_f32_16x16x4_f32 of a chunk take element e of both operands, so together they cover the chunk's 16 k.
// 4 LDS stages of 32 fragment slots (16 A row blocks + up to 13 B row blocks + padding loads that keep every wave at
// exactly 4 loads per chunk), raw s_barrier + counted s_waitcnt vmcnt: three chunks of loads stay in flight across
// the barriers (cdna_hip_programming.md section 5, "Pipelining across barriers").  Wave w owns rows [32w, 32w+32) of
// the tile: 2 x 13 accumulator tiles = 104 VGPRs, 104 MFMAs per chunk against 15 ds_read_b128.  The MFMA's A operand
// is the weight fragment, so a lane ends up with 4 consecutive output columns of one row: 16-byte stores.
#include <stdlib.h>

#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

using namespace mmqg;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxNB = 13;          // B fragments (16 columns each) per tile: BN <= 208

// W waves per workgroup, each owning 32 rows of the tile (BM = 32 W).  Per chunk a stage holds 2W A fragments + NB B
// fragments; every wave issues the same number of loads per chunk (padding loads re-read B fragment 0 into a slot nobody
// reads) so that the counted vmcnt is a compile-time constant.
template <int NB, int W>
struct Geo {
    static constexpr int kBM = 32 * W;
    static constexpr int kThreads = 64 * W;
    static constexpr int kLoads = (2 * W + NB + W - 1) / W;            // per wave and chunk
    static constexpr int kSlots = kLoads * W;
    static constexpr int kStages = W == 8 ? 4 : 3;
    static constexpr int kLdsBytes = kStages * kSlots * 1024;
};

struct NtArgs {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    const float* bias; const float* bias2;
    float* C; int ldc;
    int tiles_n, tiles;             // column tiles, total tiles
    float4* stats;                  // nullable [M][tiles_n]: per row and column tile {max, sum exp(x - max), argmax index bits, 0}
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NB, int W, bool STATS>
__global__ __launch_bounds__(64 * W, 2) void gemm_nt_tile_kernel(NtArgs p) {
    using G = Geo<NB, W>;
    constexpr int kBM = G::kBM, kStages = G::kStages, kSlots = G::kSlots, kLoads = G::kLoads;
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];      // [stage][slot][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    const int nch = p.K / 16;
    constexpr int BN = NB * 16;

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
        const int m0 = tm * kBM, n0 = tn * BN;
        // this wave's fragment slots of every chunk: slot = kLoads*wave + q; slots 0..2W-1 = A row blocks, then the NB B
        // row blocks, anything beyond re-loads B block 0 into a padding slot nobody reads (constant loads per wave)
        const float* src[kLoads];
#pragma unroll
        for (int q = 0; q < kLoads; ++q) {
            const int slot = kLoads * wave + q;
            if (slot < 2 * W) src[q] = p.A + (int64_t)min(m0 + slot * 16 + fi, p.M - 1) * p.lda + 4 * kq;
            else {
                const int nb = slot - 2 * W < NB ? slot - 2 * W : 0;
                src[q] = p.B + (int64_t)min(n0 + nb * 16 + fi, p.N - 1) * p.ldb + 4 * kq;
            }
        }
        auto issue = [&](int c) {                                      // chunk c -> stage c % kStages (c clamped: padding)
            const int cc = min(c, nch - 1);
            f32x4* st = lds + (c % kStages) * kSlots * 64;
#pragma unroll
            for (int q = 0; q < kLoads; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(src[q] + 16 * cc), (lptr_t)(st + (kLoads * wave + q) * 64), 16, 0, 0);
        };
        f32x4 acc[2][NB];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[r][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int c = 0; c < kStages - 1; ++c) issue(c);
        for (int c = 0; c < nch; ++c) {
            // chunk c has landed once at most the (kStages - 2) younger chunks' loads are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads * (kStages - 2)) : "memory");
            __builtin_amdgcn_s_barrier();                              // ... for every wave; and chunk c-1 is consumed
            asm volatile("" ::: "memory");
            issue(c + kStages - 1);                                    // into the stage chunk c-1 just left
            const f32x4* st = lds + (c % kStages) * kSlots * 64;
            const f32x4 a0 = st[(2 * wave) * 64 + lane], a1 = st[(2 * wave + 1) * 64 + lane];
            // all weight fragments of the chunk are requested from LDS up front (LDS returns in order: the MFMAs of fragment
            // nb only wait for reads 0..nb); read-then-use per fragment left every group of 8 MFMAs behind an LDS round trip
            f32x4 bf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = st[(2 * W + nb) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 b = bf[nb];
                acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a0.x, acc[0][nb], 0, 0, 0);
                acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a1.x, acc[1][nb], 0, 0, 0);
                acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a0.y, acc[0][nb], 0, 0, 0);
                acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a1.y, acc[1][nb], 0, 0, 0);
                acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a0.z, acc[0][nb], 0, 0, 0);
                acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a1.z, acc[1][nb], 0, 0, 0);
                acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a0.w, acc[0][nb], 0, 0, 0);
                acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a1.w, acc[1][nb], 0, 0, 0);
            }
        }
        // epilogue.  D[i][j] of the 16x16 MFMA: A-operand index i = weight row (output column), B-operand index j =
        // activation row: lane holds row j = lane & 15 and the 4 consecutive columns 4*(lane >> 4) + {0..3}
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = m0 + (2 * wave + r) * 16 + fi;
            const bool row_ok = m < p.M;
            float* crow = p.C + (int64_t)min(m, p.M - 1) * p.ldc;
            // bias first (in place), so that the row statistics below see the finished logits
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + nb * 16 + 4 * kq;
                if (n + 3 < p.N) {
                    if (p.bias) acc[r][nb] += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.bias2) acc[r][nb] += *reinterpret_cast<const f32x4*>(p.bias2 + n);
                } else {
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.N) acc[r][nb][e] += (p.bias ? p.bias[n + e] : 0.f) + (p.bias2 ? p.bias2[n + e] : 0.f);
                        else if (STATS) acc[r][nb][e] = -INFINITY;       // columns past N: out of the statistics, never stored
                    }
                }
            }
            if (STATS) {
                // cross-entropy statistics of this row over the tile's columns (decoder.py:106 + train.py:174: the
                // log-softmax needs max and sum-exp of the whole row; the loss kernel combines the column tiles'
                // partial results instead of sweeping the logits twice more).  A row's columns sit in this lane and
                // in the lanes +16, +32, +48.
                float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = n0 + nb * 16 + 4 * kq + e;
                        const float x = acc[r][nb][e];
                        if (x > best || (x == best && n < bi)) { best = x; bi = n; }      // first maximum wins
                    }
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float ob = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                float sum = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sum += expf(acc[r][nb][e] - best);
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                if (kq == 0 && row_ok) p.stats[(int64_t)m * p.tiles_n + tn] = make_float4(best, sum, __int_as_float(bi), 0.f);
            }
            if (!row_ok) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + nb * 16 + 4 * kq;
                if (n >= p.N) continue;
                const f32x4 v = acc[r][nb];
                if (n + 3 < p.N) {
                    *reinterpret_cast<f32x4*>(crow + n) = v;        // (non-temporal stores measured 3% slower)
                } else {
                    for (int e = 0; e < 4 && n + e < p.N; ++e) crow[n + e] = v[e];
                }
            }
        }
        // the padding loads of the last chunks and the stores are drained, and every wave is done reading the
        // stages, before the next tile's prologue writes them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

int device_cus() {
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        return p.multiProcessorCount;
    }();
    return n;
}

template <int NB, int W, bool STATS>
int launch_nt2(const NtArgs& a, int grid, hipStream_t s) {
    static int attr = 0;
    const int lds_bytes = Geo<NB, W>::kLdsBytes;
    if (attr == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_tile_kernel<NB, W, STATS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) (void)hipGetLastError();
        attr = e == hipSuccess ? 1 : -1;
    }
    if (attr < 0) return 1;
    hipLaunchKernelGGL((gemm_nt_tile_kernel<NB, W, STATS>), dim3(grid), dim3(64 * W), (size_t)lds_bytes, s, a);
    return check_launch("gemm_nt_tile");
}

template <int NB, int W>
int launch_nt(const NtArgs& a, int grid, hipStream_t s) {
    return a.stats ? launch_nt2<NB, W, true>(a, grid, s) : launch_nt2<NB, W, false>(a, grid, s);
}

}  // namespace

namespace mmqg {

// 0 = launched, 1 = shape / operands not taken (the caller uses the generic tiled GEMM), < 0 = error
int64_t gemm_nt_stats_bytes(int M, int N) { return (int64_t)M * ceil_div(N, 128) * 16; }

int gemm_nt_tile(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias,
                 const float* bias2, float* C, int ldc, hipStream_t s, float* stats, int64_t stats_bytes, int* stats_tiles) {
    if (stats_tiles) *stats_tiles = 0;
    static const bool off = [] { const char* e = getenv("MMQG_NO_NT_TILE"); return e && atoi(e) != 0; }();
    if (off) return 1;
    const int cus = device_cus();
    if (cus < 64 || K < 64 || K % 16 != 0 || lda % 4 != 0 || ldb % 4 != 0 || ldc % 4 != 0) return 1;
    if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (bias && !aligned16(bias)) || (bias2 && !aligned16(bias2))) return 1;
    // worth it only when the output fills the chip with tiles of this size: at least ~0.9 tiles of 256 x 128 per CU
    static const int W = [] { const char* e = getenv("MMQG_NT_WAVES"); return e && atoi(e) == 4 ? 4 : 8; }();
    const int kBM = 32 * W, slots = cus * (8 / W);                    // W = 4: two workgroups share a CU
    if ((int64_t)M * N < (int64_t)slots * kBM * 112) return 1;
    // column-tile width: the multiple of 16 (128 .. 208) that wastes the least of the chip, counting whole rounds of
    // `cus` tiles and the padding of edge tiles
    const int tiles_m = ceil_div(M, kBM);
    int best_nb = 0;
    double best_eff = 0.0;
    for (int nb = 8; nb <= kMaxNB; ++nb) {
        if (nb == 9 || nb == 11) continue;                            // instantiated: 8, 10, 12, 13
        const int tn = ceil_div(N, nb * 16);
        const int64_t tiles = (int64_t)tiles_m * tn;
        const int64_t rounds = (tiles + slots - 1) / slots;
        const double eff = (double)M * N / ((double)rounds * slots * kBM * nb * 16);
        if (eff > best_eff) { best_eff = eff; best_nb = nb; }
    }
    if (best_nb == 0 || best_eff < 0.70) return 1;
    NtArgs a{M, N, K, A, lda, B, ldb, bias, bias2, C, ldc, ceil_div(N, best_nb * 16), 0, nullptr};
    a.tiles = tiles_m * a.tiles_n;
    if (stats && stats_tiles && aligned16(stats) && stats_bytes >= (int64_t)M * a.tiles_n * 16) {
        a.stats = reinterpret_cast<float4*>(stats);
        *stats_tiles = a.tiles_n;
    }
    const int grid = std::min(a.tiles, slots);
    int rc;
    if (W == 8) {
        switch (best_nb) {
            case 8: rc = launch_nt<8, 8>(a, grid, s); break;
            case 10: rc = launch_nt<10, 8>(a, grid, s); break;
            case 12: rc = launch_nt<12, 8>(a, grid, s); break;
            default: rc = launch_nt<13, 8>(a, grid, s); break;
        }
    } else {
        switch (best_nb) {
            case 8: rc = launch_nt<8, 4>(a, grid, s); break;
            case 10: rc = launch_nt<10, 4>(a, grid, s); break;
            case 12: rc = launch_nt<12, 4>(a, grid, s); break;
            default: rc = launch_nt<13, 4>(a, grid, s); break;
        }
    }
    if (rc != 0 && stats_tiles) *stats_tiles = 0;
    return rc;
}

}  // namespace mmqg
