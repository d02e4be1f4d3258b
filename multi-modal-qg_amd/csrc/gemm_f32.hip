// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit an fmaf chain, so results stay inside the 1e-4 parity budget).
//
//   C[M,N] = (beta ? C : 0) + bias[N] + A(M,K) * B(K,N)  [+ A2(M,K2) * B2(K2,N)]
//
// Operand layouts (no copies, the caller's leading dimensions are used as they are):
//   A k-major : element (m,k) at A[m*lda + k]   (activations x[B,in])
//   A m-major : element (m,k) at A[k*lda + m]   (dgates^T for weight gradients)
//   B k-major : element (k,n) at B[n*ldb + k]   (torch Linear/LSTM weight [out,in] -> x*W^T)
//   B n-major : element (k,n) at B[k*ldb + n]   (dgates*W, dgates^T*x)
// The optional second operand pair shares the layouts and extends the K loop, which is how
// the LSTM gate pre-activations x*W_ih^T + h*W_hh^T are produced by one launch.
//
// Tiling: 256 threads = 4 wavefronts (2x2).  Two shapes:
//   BIG   128x128x{16,32}, each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs)
//   SMALL  64x64x32,  each wave 32x32 = 1 MFMA tile; used with split-K (atomic f32 adds)
//          for the batch-sized (M<=64) products inside the recurrent loops so that more
//          than a handful of the 256 CUs get work.
// Interior tiles of 16-byte-aligned operands run a body whose loaders have no bounds logic at all
// (plain global_load_dwordx4 per staged float4); only edge tiles and the K tail pay for the checks.
// That alone took 4096^3 from 104 to 132 TFLOP/s (0.66 -> 0.84 of the fp32 MFMA peak).
// LDS holds both operands k-major ([k][m] and [k][n], two stages) so an MFMA operand read is
// one conflict-free ds_read_b32 per lane; global loads are 16 B per lane and are issued for
// tile t+1 before the MFMAs of tile t, then written to the other LDS stage (one barrier per tile).
#include <stdlib.h>

#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    int M, N, K, K2;
    const float* A;  int lda;
    const float* B;  int ldb;
    const float* A2; int lda2;
    const float* B2; int ldb2;
    const float* bias; const float* bias2;
    float* C; int ldc;
    int beta;          // 0: overwrite, 1: accumulate
    int split_k;       // >1: every z-slice adds its partial product with f32 atomics
    int vec_a, vec_b, vec_a2, vec_b2;   // operand may be read with aligned 16-B loads
    int fast;                           // all operands 16-byte aligned: interior tiles load without bounds logic
};

// One operand tile of ROWS (m or n) x BK (k), staged through registers.
template <int ROWS, int BK, bool KMAJOR>
struct TileLoader {
    static constexpr int NV = ROWS * BK / 4 / 256;   // float4 per thread
    static_assert(NV >= 1, "tile too small");
    // LDS row length ([k][ROWS + pad]).  k-major sources are transposed on the way in with
    // scalar ds_write_b32; the pad makes those writes hit 32 distinct banks per half-wave.
    static constexpr int F4_PER_ROW = BK / 4;
    static constexpr int LD = KMAJOR ? ROWS + (F4_PER_ROW == 8 ? 1 : 2) : ROWS + 4;

    float4 v[NV];

    // CHECK = false: the whole tile is inside the operand and 16-byte aligned (interior tiles of aligned
    // operands, the common case) -> NV plain 16-byte loads, no bounds logic in the main loop
    template <bool CHECK>
    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int K,
                                         bool vec) {
        const int t = threadIdx.x;
        if constexpr (KMAJOR) {
            constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
            const int kq = t % F4_PER_ROW, r = t / F4_PER_ROW;
            if constexpr (!CHECK) {
#pragma unroll
                for (int p = 0; p < NV; ++p) {
                    const float4 x = *reinterpret_cast<const float4*>(P + (int64_t)(row0 + p * ROWS_PER_PASS + r) * ld + k0 + 4 * kq);
                    v[p] = x;
                }
            } else {
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int row = row0 + p * ROWS_PER_PASS + r;
                const int k = k0 + 4 * kq;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < nrows) {
                    const float* src = P + (int64_t)row * ld + k;
                    if (vec && k + 3 < K) {
                        x = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (k + 0 < K) x.x = src[0];
                        if (k + 1 < K) x.y = src[1];
                        if (k + 2 < K) x.z = src[2];
                        if (k + 3 < K) x.w = src[3];
                    }
                }
                v[p] = x;
            }
            }
        } else {
            constexpr int F4_PER_K = ROWS / 4;
            constexpr int K_PER_PASS = 256 / F4_PER_K;
            const int c4 = t % F4_PER_K, kr = t / F4_PER_K;
            if constexpr (!CHECK) {
#pragma unroll
                for (int p = 0; p < NV; ++p) {
                    const float4 x = *reinterpret_cast<const float4*>(P + (int64_t)(k0 + p * K_PER_PASS + kr) * ld + row0 + 4 * c4);
                    v[p] = x;
                }
            } else {
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int k = k0 + p * K_PER_PASS + kr;
                const int row = row0 + 4 * c4;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K) {
                    const float* src = P + (int64_t)k * ld + row;
                    if (vec && row + 3 < nrows) {
                        x = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (row + 0 < nrows) x.x = src[0];
                        if (row + 1 < nrows) x.y = src[1];
                        if (row + 2 < nrows) x.z = src[2];
                        if (row + 3 < nrows) x.w = src[3];
                    }
                }
                v[p] = x;
            }
            }
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ S) const {
        const int t = threadIdx.x;
        if constexpr (KMAJOR) {
            constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
            const int kq = t % F4_PER_ROW, r = t / F4_PER_ROW;
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int m = p * ROWS_PER_PASS + r;
                S[(4 * kq + 0) * LD + m] = v[p].x;
                S[(4 * kq + 1) * LD + m] = v[p].y;
                S[(4 * kq + 2) * LD + m] = v[p].z;
                S[(4 * kq + 3) * LD + m] = v[p].w;
            }
        } else {
            constexpr int F4_PER_K = ROWS / 4;
            constexpr int K_PER_PASS = 256 / F4_PER_K;
            const int c4 = t % F4_PER_K, kr = t / F4_PER_K;
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                *reinterpret_cast<float4*>(&S[(p * K_PER_PASS + kr) * LD + 4 * c4]) = v[p];
            }
        }
    }
};

template <int BM, int BN, int BK, bool A_K, bool B_K, bool CHECK>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, float* smem, int tile_x, int tile_y, int kslice) {
    using LA = TileLoader<BM, BK, A_K>;
    using LB = TileLoader<BN, BK, B_K>;
    constexpr int WTM = BM / 2, WTN = BN / 2;      // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;    // MFMA tiles per wave
    constexpr int A_ELEMS = BK * LA::LD, B_ELEMS = BK * LB::LD;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = tile_y * BM, n0 = tile_x * BN;

    const int nk1 = (p.K + BK - 1) / BK;
    const int nk2 = p.A2 ? (p.K2 + BK - 1) / BK : 0;
    const int nk = nk1 + nk2;
    int kt_begin = 0, kt_end = nk;
    if (p.split_k > 1) {
        const int per = (nk + p.split_k - 1) / p.split_k;
        kt_begin = kslice * per;
        kt_end = min(nk, kt_begin + per);
        if (kt_begin >= kt_end) return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LA la; LB lb;
    // operand descriptors as plain locals: a lambda that captured the argument struct by reference made
    // hipcc keep a copy of it in scratch
    const float* const A1 = p.A; const float* const B1 = p.B; const float* const A2 = p.A2; const float* const B2 = p.B2;
    const int lda1 = p.lda, ldb1 = p.ldb, lda2 = p.lda2, ldb2 = p.ldb2, M = p.M, N = p.N, K1 = p.K, K2 = p.K2;
    const bool va1 = p.vec_a, vb1 = p.vec_b, va2 = p.vec_a2, vb2 = p.vec_b2;
#define MMQG_FETCH(kt_)                                                                    \
    do {                                                                                   \
        const int kt__ = (kt_);                                                            \
        if (kt__ < nk1) {                                                                  \
            if (!CHECK && (kt__ + 1) * BK <= K1) {        /* k-tile inside K: no checks */ \
                la.template load<false>(A1, lda1, m0, M, kt__ * BK, K1, va1);              \
                lb.template load<false>(B1, ldb1, n0, N, kt__ * BK, K1, vb1);              \
            } else {                                                                       \
                la.template load<true>(A1, lda1, m0, M, kt__ * BK, K1, va1);               \
                lb.template load<true>(B1, ldb1, n0, N, kt__ * BK, K1, vb1);               \
            }                                                                              \
        } else {                                                                           \
            if (!CHECK && (kt__ - nk1 + 1) * BK <= K2) {                                   \
                la.template load<false>(A2, lda2, m0, M, (kt__ - nk1) * BK, K2, va2);      \
                lb.template load<false>(B2, ldb2, n0, N, (kt__ - nk1) * BK, K2, vb2);      \
            } else {                                                                       \
                la.template load<true>(A2, lda2, m0, M, (kt__ - nk1) * BK, K2, va2);       \
                lb.template load<true>(B2, ldb2, n0, N, (kt__ - nk1) * BK, K2, vb2);       \
            }                                                                              \
        }                                                                                  \
    } while (0)

    MMQG_FETCH(kt_begin);
    la.store(smem);
    lb.store(smem + A_ELEMS);
    __syncthreads();

    const int half = lane >> 5, l31 = lane & 31;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const float* As = smem + cur * (A_ELEMS + B_ELEMS);
        const float* Bs = As + A_ELEMS;
        const bool more = kt + 1 < kt_end;
        if (more) MMQG_FETCH(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(kk + half) * LA::LD + wr * WTM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + half) * LB::LD + wc * WTN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* nxt = smem + (cur ^ 1) * (A_ELEMS + B_ELEMS);
            la.store(nxt);
            lb.store(nxt + A_ELEMS);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool lead = (p.split_k <= 1) || (kslice == 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * WTN + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = lead ? ((p.bias ? p.bias[col] : 0.f) + (p.bias2 ? p.bias2[col] : 0.f)) : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= p.M) continue;
                float* dst = p.C + (int64_t)row * p.ldc + col;
                const float val = acc[i][j][r] + bv;
                if (p.split_k > 1) {
                    atomicAdd(dst, val);
                } else {
                    *dst = p.beta ? (*dst + val) : val;
                }
            }
        }
    }
}

#undef MMQG_FETCH

template <int BM, int BN, int BK, bool A_K, bool B_K>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    using LA = TileLoader<BM, BK, A_K>;
    using LB = TileLoader<BN, BK, B_K>;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LA::LD + LB::LD)];
    // interior tiles of aligned operands (p.fast, decided on the host) skip every bounds check of the loaders
    const bool interior = p.fast && (int)(blockIdx.y + 1) * BM <= p.M && (int)(blockIdx.x + 1) * BN <= p.N;
    if (interior) gemm_body<BM, BN, BK, A_K, B_K, false>(p, smem, blockIdx.x, blockIdx.y, blockIdx.z);
    else gemm_body<BM, BN, BK, A_K, B_K, true>(p, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent products of one layout in ONE launch (weight gradients of a layer stack): grid.z =
// problem x k-slice, the x/y extent is the largest problem's tile grid, every problem accumulates (beta = 1)
// so k-slices simply add atomically.
constexpr int kMaxGroup = 10;
struct GemmBatch { GemmArgs p[kMaxGroup]; int split; };

template <int BM, int BN, int BK, bool A_K, bool B_K>
__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(GemmBatch b) {
    using LA = TileLoader<BM, BK, A_K>;
    using LB = TileLoader<BN, BK, B_K>;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LA::LD + LB::LD)];
    const int prob = blockIdx.z / b.split, kslice = blockIdx.z % b.split;
    const GemmArgs& p = b.p[prob];
    if ((int)blockIdx.y * BM >= p.M || (int)blockIdx.x * BN >= p.N) return;
    const bool interior = p.fast && (int)(blockIdx.y + 1) * BM <= p.M && (int)(blockIdx.x + 1) * BN <= p.N;
    if (interior) gemm_body<BM, BN, BK, A_K, B_K, false>(p, smem, blockIdx.x, blockIdx.y, kslice);
    else gemm_body<BM, BN, BK, A_K, B_K, true>(p, smem, blockIdx.x, blockIdx.y, kslice);
}

template <int BM, int BN, int BK>
int launch(const GemmArgs& a, int a_layout, int b_layout, hipStream_t s) {
    dim3 grid(mmqg::ceil_div(a.N, BN), mmqg::ceil_div(a.M, BM), a.split_k > 1 ? a.split_k : 1);
    dim3 block(256);
    if (a_layout == MMQG_K_MAJOR && b_layout == MMQG_K_MAJOR)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, true, true>), grid, block, 0, s, a);
    else if (a_layout == MMQG_K_MAJOR && b_layout == MMQG_MN_MAJOR)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, true, false>), grid, block, 0, s, a);
    else if (a_layout == MMQG_MN_MAJOR && b_layout == MMQG_MN_MAJOR)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, false, false>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, false, true>), grid, block, 0, s, a);
    return mmqg::check_launch("gemm_f32");
}

// K split of a 128x128-tiled product: the number of rounds the workgroups need on the chip's resident
// slots times the length of one workgroup's k-loop (plus a fixed per-workgroup cost in k-tile units:
// pipeline fill, epilogue, atomics), minimised over 1, 2, 4, ... — e.g. 576 tiles x 80 k-tiles on 768 slots:
// split 2 = 2 rounds x 46, split 4 = 3 rounds x 26, so 4 wins although 2 already "fills" the chip.
inline int pick_split(int64_t tiles, int nk, int slots, int fixed, int max_split, int min_kt) {
    int best = 1;
    int64_t best_cost = ((tiles + slots - 1) / slots) * (int64_t)(nk + fixed);
    for (int sp = 2; sp <= max_split && nk / sp >= min_kt; sp *= 2) {
        const int64_t cost = ((tiles * sp + slots - 1) / slots) * (int64_t)((nk + sp - 1) / sp + fixed);
        if (cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}

inline bool can_vec(const float* p, int ld) { return p && mmqg::aligned16(p) && (ld % 4 == 0); }

// zero a [M][N] block with leading dimension ldc (split-K partial sums are added atomically)
__global__ __launch_bounds__(256) void zero_block_kernel(float* C, int M, int N, int ldc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    C[(i / N) * ldc + (i % N)] = 0.f;
}

// compact, 16-byte aligned block: four floats per thread, grid-stride
__global__ __launch_bounds__(256) void zero_flat4_kernel(float4* C, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) C[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

namespace mmqg {

int gemm_f32(int a_layout, int b_layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
             const float* A2, int lda2, const float* B2, int ldb2, int K2, const float* bias, const float* bias2,
             int beta, float* C, int ldc, int split_k, hipStream_t s) {
    MMQG_REQUIRE(M >= 0 && N >= 0 && K >= 1, "gemm_f32: bad dimension (need M,N >= 0 and K >= 1)");
    if (M == 0 || N == 0) return 0;
    MMQG_REQUIRE(A && B && C, "gemm_f32: null operand");
    MMQG_REQUIRE((A2 == nullptr) == (B2 == nullptr), "gemm_f32: second operand pair must be both set or both null");
    MMQG_REQUIRE(a_layout == MMQG_K_MAJOR || a_layout == MMQG_MN_MAJOR, "gemm_f32: bad a_layout");
    MMQG_REQUIRE(b_layout == MMQG_K_MAJOR || b_layout == MMQG_MN_MAJOR, "gemm_f32: bad b_layout");
    MMQG_REQUIRE(lda >= (a_layout == MMQG_K_MAJOR ? K : M), "gemm_f32: lda too small");
    MMQG_REQUIRE(ldb >= (b_layout == MMQG_K_MAJOR ? K : N), "gemm_f32: ldb too small");
    MMQG_REQUIRE(ldc >= N, "gemm_f32: ldc too small");
    // large k-major x k-major products with an overwritten output (the vocabulary projection): one big tile per CU
    if (a_layout == MMQG_K_MAJOR && b_layout == MMQG_K_MAJOR && !A2 && !beta && split_k <= 1) {
        const int rc = gemm_nt_tile(M, N, K, A, lda, B, ldb, bias, bias2, C, ldc, s);
        if (rc <= 0) return rc;
    }
    // large single products: the split-bf16 kernel (fp32-exact operands on the bf16 matrix cores, gemm_x3.hip)
    if (!A2 && !(a_layout == MMQG_MN_MAJOR && b_layout == MMQG_K_MAJOR) && gemm_x3_wants(M, N, K)) {
        const GemmProblem q{M, N, K, A, lda, B, ldb, C, ldc, beta ? 1 : 0};
        const int rc = gemm_x3_grouped(a_layout, b_layout, &q, &bias, &bias2, 1, s);
        if (rc <= 0) return rc;
    }
    GemmArgs a;
    a.M = M; a.N = N; a.K = K; a.K2 = A2 ? K2 : 0;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb;
    a.A2 = A2; a.lda2 = lda2; a.B2 = B2; a.ldb2 = ldb2;
    a.bias = bias; a.bias2 = bias2; a.C = C; a.ldc = ldc; a.beta = beta ? 1 : 0;
    a.vec_a = can_vec(A, lda); a.vec_b = can_vec(B, ldb);
    a.vec_a2 = can_vec(A2, lda2); a.vec_b2 = can_vec(B2, ldb2);

    // Tile shape and K split.  The 128x128 tile reuses each operand byte 4x more than the 64x64 one,
    // so it is preferred whenever both extents reach 128; K is then split (f32 atomics) until the grid
    // has at least ~2 workgroups per CU, keeping >= 8 k-tiles per slice.  MMQG_GEMM_HEUR=1 restores the
    // round-1 rule (big tile only for >= 96 tiles, split-K only on the small tile).
    static const int heur = env_int("MMQG_GEMM_HEUR", 2);
    // k-tile of the 128x128 shape: 32 (one barrier per 64 MFMAs, 2 workgroups per CU) suits the k-major
    // x k-major products with short K (hoisted input products, vocabulary projection); 16 (33 KB of LDS, 3
    // workgroups per CU) measured faster for the n-major-B products (data / weight gradients, mostly split-K).
    // MMQG_GEMM_BIG_BK=16|32 forces one of them.
    static const int big_bk_env = env_int("MMQG_GEMM_BIG_BK", 0);
    const int big_bk = big_bk_env ? big_bk_env : (b_layout == MMQG_MN_MAJOR ? 16 : 32);
    const int64_t big_tiles = (int64_t)ceil_div(M, 128) * ceil_div(N, 128);
    bool small;
    if (heur == 1) small = (M <= 64) || (N <= 64) || big_tiles < 96;
    else small = (M <= 64) || (N <= 64) || (M < 128 && N < 256) || (N < 128 && M < 256);
    const int bk = small ? 32 : big_bk;
    const int nk = ceil_div(K, bk) + (A2 ? ceil_div(K2, bk) : 0);
    if (split_k < 0) {
        split_k = 1;
        if (small) {      // aim for ~384 workgroups, at least 4 k-tiles per slice
            const int64_t tiles = (int64_t)ceil_div(M, 64) * ceil_div(N, 64);
            while (tiles * split_k < 384 && nk / (split_k * 2) >= 4 && split_k < 16) split_k *= 2;
        } else if (heur == 3) {     // cost model (rounds x k-loop length), as for the grouped launches
            split_k = pick_split(big_tiles, nk, bk == 16 ? 768 : 512, bk == 16 ? 6 : 3, 16, 8);
        } else if (heur != 1) {
            while (big_tiles * split_k < 448 && nk / (split_k * 2) >= 8 && split_k < 16) split_k *= 2;
        }
    }
    static const int max_split = env_int("MMQG_MAX_SPLITK", 16);      // diagnostics: 1 disables split-K
    // diagnostics: 1 = hipMemsetAsync (races with neighbouring kernel nodes under hipGraph replay on
    // ROCm 7.2 / gfx950: NaNs after a few hundred replays; the default is the zeroing kernel above)
    static const int memset_api = env_int("MMQG_MEMSET_API", 0);
    if (split_k > max_split) split_k = max_split;
    if (split_k > nk) split_k = nk > 0 ? nk : 1;
    a.split_k = split_k;
    static const int no_fast = env_int("MMQG_GEMM_NO_FAST", 0);
    a.fast = !no_fast && a.vec_a && a.vec_b && (!A2 || (a.vec_a2 && a.vec_b2));
    if (split_k > 1 && !beta) {
        // partial products are added atomically, so the destination must start from zero
        if (memset_api && ldc == N) {
            hipError_t e = hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N, s);
            MMQG_REQUIRE(e == hipSuccess, "gemm_f32: memset failed: %s", hipGetErrorString(e));
        } else if (ldc == N && ((int64_t)M * N) % 4 == 0 && aligned16(C)) {
            const int64_t n4 = (int64_t)M * N / 4;
            hipLaunchKernelGGL(zero_flat4_kernel, dim3((unsigned)std::min<int64_t>(mmqg::ceil_div64(n4, 256), 2048)), dim3(256), 0, s,
                               reinterpret_cast<float4*>(C), n4);
            MMQG_TRY(mmqg::check_launch("gemm_f32 zero"));
        } else {
            hipLaunchKernelGGL(zero_block_kernel, dim3((unsigned)mmqg::ceil_div64((int64_t)M * N, 256)), dim3(256), 0, s,
                               C, M, N, ldc);
            MMQG_TRY(mmqg::check_launch("gemm_f32 zero"));
        }
    }
    if (small) return launch<64, 64, 32>(a, a_layout, b_layout, s);
    static const int tile_env = env_int("MMQG_GEMM_TILE", 0);      // experiments: 1 = 64x128x32, 2 = 128x64x32, 3 = 64x128x16
    if (tile_env == 1) return launch<64, 128, 32>(a, a_layout, b_layout, s);
    if (tile_env == 2) return launch<128, 64, 32>(a, a_layout, b_layout, s);
    if (tile_env == 3) return launch<64, 128, 16>(a, a_layout, b_layout, s);
    if (big_bk == 32) return launch<128, 128, 32>(a, a_layout, b_layout, s);
    return launch<128, 128, 16>(a, a_layout, b_layout, s);
}

int gemm_f32_wgrad_group(const GemmProblem* probs, float* const* colsum, float* const* colsum2, int n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && (n == 0 || probs), "gemm_f32_wgrad_group: bad arguments");
    constexpr int cap = 2 * MMQG_MAX_LAYERS + 4;
    GemmProblem big[cap], rest[cap];
    float* cs1[cap]; float* cs2[cap];
    int src[cap];                       // index in probs of big[i]
    bool took[cap] = {};                // probs[i] goes to the split-bf16 launch (with its column sums)
    int nb = 0, nr = 0;
    bool fused = n <= cap;
    for (int i = 0; i < n && fused; ++i) {
        const GemmProblem& q = probs[i];
        if (q.A && q.B && q.C && gemm_x3_wants(q.M, q.N, q.K)) {
            big[nb] = q; cs1[nb] = colsum ? colsum[i] : nullptr; cs2[nb] = colsum2 ? colsum2[i] : nullptr; src[nb] = i; took[i] = true; ++nb;
        } else {
            rest[nr++] = q;
        }
    }
    // OPT-IN (MMQG_X3_PEEL=1).  One 256 x 128 tile per workgroup and CU: a group whose tiles just exceed the CU count pays a
    // whole second round for the excess (the decoder's group at config 2: 264 tiles on 256 CUs — 8 of them from dW_attn).
    // Peeling the smallest products back to the fp32 kernels (their column sums to a sweep of their own) brings the group
    // under the CU count.  Measured (round 4, one box): the decoder's backward phase alone 1,424 -> 1,359 us, but the step
    // 4.20-4.22 -> 4.23-4.24 ms: in the step the group runs beside the text encoder's weight gradients, which use the CUs
    // the thin second round leaves idle, and the peeled products cost two more launches.
    if (fused && nb > 1) {
        const int slots = gemm_x3_slots();
        auto tiles_of = [](const GemmProblem& q) { return ceil_div(q.M, 256) * ceil_div(q.N, 128); };
        int total = 0;
        for (int i = 0; i < nb; ++i) total += tiles_of(big[i]);
        const int rounds = slots > 0 ? ceil_div(total, slots) : 1;
        const int excess = slots > 0 ? total - (rounds - 1) * slots : 0;          // tiles in the last round
        static const bool peel = [] { const char* e = getenv("MMQG_X3_PEEL"); return e && atoi(e) != 0; }();
        if (rounds > 1 && excess * 8 <= total && peel) {
            int peeled = 0;
            while (peeled < excess && nb > 1) {
                int best = 0;
                for (int i = 1; i < nb; ++i)
                    if (tiles_of(big[i]) < tiles_of(big[best])) best = i;
                if (peeled + tiles_of(big[best]) > 2 * excess) break;
                peeled += tiles_of(big[best]);
                rest[nr++] = big[best];
                took[src[best]] = false;
                for (int i = best; i + 1 < nb; ++i) { big[i] = big[i + 1]; cs1[i] = cs1[i + 1]; cs2[i] = cs2[i + 1]; src[i] = src[i + 1]; }
                --nb;
            }
        }
    }
    if (fused && nb > 0) {
        const int rc = gemm_x3_grouped(MMQG_MN_MAJOR, MMQG_MN_MAJOR, big, nullptr, nullptr, nb, s, cs1, cs2);
        if (rc < 0) return rc;
        fused = rc == 0;
    } else {
        fused = false;
    }
    // column sums the fused launch did not take: one sweep over A each (A_i is [K][M], lda)
    for (int i = 0; i < n; ++i) {
        if (!colsum || !colsum[i]) continue;
        const GemmProblem& q = probs[i];
        const bool taken = fused && i < cap && took[i];
        if (!taken) MMQG_TRY(colsum_add2(q.A, q.lda, q.K, q.M, colsum[i], colsum2 ? colsum2[i] : nullptr, s));
    }
    if (fused) return nr > 0 ? gemm_f32_grouped(MMQG_MN_MAJOR, MMQG_MN_MAJOR, rest, nr, s) : 0;
    return gemm_f32_grouped(MMQG_MN_MAJOR, MMQG_MN_MAJOR, probs, n, s);
}

int gemm_f32_grouped(int a_layout, int b_layout, const GemmProblem* probs, int n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && (n == 0 || probs), "gemm_f32_grouped: bad arguments");
    static const int no_group = env_int("MMQG_GEMM_NO_GROUP", 0);
    // one launch pays off for 128x128-tileable accumulating products; anything else goes one by one
    bool ok = !no_group && n >= 2 && a_layout == MMQG_MN_MAJOR && b_layout == MMQG_MN_MAJOR;
    for (int i = 0; i < n && ok; ++i)
        ok = probs[i].beta == 1 && probs[i].M >= 128 && probs[i].N >= 128 && probs[i].K >= 1 && probs[i].A && probs[i].B && probs[i].C;
    // the large products of a group go to the split-bf16 kernel (every layout pair it has, any beta) in one launch;
    // what is left (short K, narrow outputs) continues below
    if (n >= 1 && !(a_layout == MMQG_MN_MAJOR && b_layout == MMQG_K_MAJOR)) {
        GemmProblem big[2 * MMQG_MAX_LAYERS + 4], rest[2 * MMQG_MAX_LAYERS + 4];
        int nb = 0, nr = 0;
        const int cap = 2 * MMQG_MAX_LAYERS + 4;
        for (int i = 0; i < n && n <= cap; ++i) {
            const bool want = probs[i].A && probs[i].B && probs[i].C && gemm_x3_wants(probs[i].M, probs[i].N, probs[i].K);
            if (want) big[nb++] = probs[i]; else rest[nr++] = probs[i];
        }
        if (nb > 0) {
            const int rc = gemm_x3_grouped(a_layout, b_layout, big, nullptr, nullptr, nb, s);
            if (rc < 0) return rc;
            if (rc == 0) return nr > 0 ? gemm_f32_grouped(a_layout, b_layout, rest, nr, s) : 0;
        }
    }
    if (!ok) {
        for (int i = 0; i < n; ++i) {
            const GemmProblem& q = probs[i];
            MMQG_TRY(gemm_f32(a_layout, b_layout, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, nullptr, 0, nullptr, 0, 0, nullptr,
                              nullptr, q.beta, q.C, q.ldc, -1, s));
        }
        return 0;
    }
    constexpr int BK = 16;
    for (int i0 = 0; i0 < n; i0 += kMaxGroup) {
        const int ng = std::min(kMaxGroup, n - i0);
        GemmBatch b{};
        int tx = 0, ty = 0, nk_min = 1 << 30;
        int64_t tiles = 0;
        for (int i = 0; i < ng; ++i) {
            const GemmProblem& q = probs[i0 + i];
            MMQG_REQUIRE(q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N, "gemm_f32_grouped: leading dimension too small");
            GemmArgs& a = b.p[i];
            a = GemmArgs{};
            a.M = q.M; a.N = q.N; a.K = q.K; a.A = q.A; a.lda = q.lda; a.B = q.B; a.ldb = q.ldb; a.C = q.C; a.ldc = q.ldc;
            a.beta = 1;
            a.vec_a = can_vec(q.A, q.lda); a.vec_b = can_vec(q.B, q.ldb);
            a.fast = a.vec_a && a.vec_b;
            nk_min = std::min(nk_min, ceil_div(q.K, BK));
        }
        // 128x64 tiles: twice the tiles of 128x128, so fewer k-slices (atomic adds) fill the chip (+0.7% step
        // throughput); MMQG_GROUP_BN=128 restores the square tile
        static const int group_bn = env_int("MMQG_GROUP_BN", 64);
        const int bn = group_bn == 64 ? 64 : 128;
        for (int i = 0; i < ng; ++i) {
            const GemmProblem& q = probs[i0 + i];
            tx = std::max(tx, ceil_div(q.N, bn)); ty = std::max(ty, ceil_div(q.M, 128));
            tiles += (int64_t)ceil_div(q.N, bn) * ceil_div(q.M, 128);
        }
        // k-slices (atomic adds; every problem accumulates anyway): ~3 workgroups per CU, >= 8 k-tiles each
        const int split = pick_split(tiles, nk_min, bn == 64 ? 1024 : 768, 6, 8, 8);
        for (int i = 0; i < ng; ++i) b.p[i].split_k = split;
        b.split = split;
        // MMQG_GROUP_MAX_WGS=n (experiments): at most n workgroups of this GEMM per CU, enforced by asking for unused
        // dynamic LDS, so that the 8-wave kernels of a dependent chain on the other stream still find room on every CU
        static const int max_wgs = env_int("MMQG_GROUP_MAX_WGS", 0);
        const int static_lds = 2 * BK * ((128 + 4) + (bn + 4)) * 4;
        size_t pad = 0;
        if (max_wgs > 0) {
            const int per = 160 * 1024 / (max_wgs + 1) + 1024;        // too big for max_wgs + 1 of them to share a CU
            if (per > static_lds) pad = (size_t)(per - static_lds);
        }
        if (bn == 64)
            hipLaunchKernelGGL((gemm_f32_grouped_kernel<128, 64, BK, false, false>), dim3(tx, ty, ng * split), dim3(256), pad, s, b);
        else
            hipLaunchKernelGGL((gemm_f32_grouped_kernel<128, 128, BK, false, false>), dim3(tx, ty, ng * split), dim3(256), pad, s, b);
        MMQG_TRY(check_launch("gemm_f32_grouped"));
    }
    return 0;
}

}  // namespace mmqg
