// Large fp32 products on the bf16 matrix cores, at fp32 accuracy ("split" GEMM).
//
// Every fp32 operand element is split EXACTLY into three bf16 pieces x = x1 + x2 + x3 (round-to-nearest pieces of
// the running remainder: 8 + 8 + 8 significand bits, the remainders are exact in fp32), and the product a*b is taken
// as the six piece products of weight >= 2^-18:  a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2).  The dropped terms are
// <= 2^-26 |a||b| — below fp32's own rounding (2^-24) — each bf16 x bf16 product is exact in fp32 and the MFMA
// accumulates in fp32, so the result carries the same error as an fp32 GEMM (tests/test_hip_kernels.py compares both
// with float64: 2-8e-7 of max|C| for either at K = 512 .. 10000).  What it buys: v_mfma_f32_32x32x16_bf16 runs at 16x
// the rate of v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md, matrix cores: 2.5 PFLOP/s against 157 TFLOP/s dense), so
// six of them per fp32 product leave a 2.7x higher ceiling than the fp32 MFMA path — and 6 MFMAs per staged element
// make the staging (global -> registers -> split -> LDS) cheap relative to the arithmetic.
//
// Serves the step's large products in all three operand layouts (reference: nn.Linear / nn.LSTM gate matmuls and their
// autograd, model/decoder.py:64-70,104,106, model/encoder.py:54,91):
//   NT  C[M,N] = A[M,K] B[N,K]^T          activations x Linear weight      (hoisted input products, vocabulary projection)
//   NN  C[M,N] = A[M,K] B[K,N]            gradient x weight                (data gradients)
//   TN  C[M,N] = A[K,M]^T B[K,N]          gradient^T x activations         (weight gradients; grouped, accumulating)
//
// Kernel: 128 x 128 output tile per 4-wave workgroup (wave = 64 x 64 = 2 x 2 MFMA tiles of 32 x 32), 32-deep k-chunks.
// A thread stages 16 consecutive k of one operand row per chunk: k-major operands by four 16-byte loads, m-major
// operands by sixteen dword loads whose lanes run along m (256 contiguous bytes per wave-instruction) — the
// transposition happens in registers, the LDS image is the same [row][32 k] bf16 for both, three planes per operand,
// rows padded to 80 bytes (ds_read_b128 of 16 consecutive rows then covers all 64 banks).  The global loads of the
// next TWO chunks are in flight while the current chunk's 48 MFMAs per wave run; two workgroups per CU (61 KB of LDS, <= 256
// VGPRs) overlap one's split/write pass with the other's MFMAs.  A lane ends up with one output column and 16 rows,
// so the stores (or atomics) of a half-wave cover 128 contiguous bytes.  Several problems ride in one launch
// (weight-gradient groups); K can be sliced across workgroups (f32 atomics into a zeroed or accumulating C).
#include <stdlib.h>

#include <algorithm>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

using namespace mmqg;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTile = 128;          // output tile edge
constexpr int kBK = 32;             // k-chunk
constexpr int kRowBytes = 80;       // 32 bf16 + 16 bytes of padding
constexpr int kPlaneBytes = 2 * kTile * kRowBytes;     // A rows 0..127, B rows 128..255
constexpr int kLdsBytes = 3 * kPlaneBytes;             // 61,440
constexpr int kMaxProblems = 12;

struct X3Problem {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias; const float* bias2;
    int beta;                        // 1: C += product
    int split_k;                     // k slices (>= 1); > 1 adds with f32 atomics
    int tiles_n;
    int wg0;                         // first workgroup of this problem in the launch
    float4* stats;                   // ping-pong kernel, STATS variant: [M][tiles_n] {max, sum exp(x - max), argmax bits, 0}
    float* colsum; float* colsum2;   // ping-pong kernel, m-major A: out[m] += sum over k of A[k][m] (bias gradients: the
                                     // column sums of the gate gradients ride in the weight-gradient product's staging pass)
    int q0, qpt;                     // balanced launches: first k-quad (4 chunks = 128 k) of this problem in the launch's
                                     // list of (tile, k-quad) units, quads per tile
};

struct X3Batch {
    int n;
    int total_q;                     // balanced launches: k-quads of all tiles of all problems
    int dword_epilogue;              // MMQG_X3_DWORD_EPILOGUE=1 (A/B): the lane-per-column epilogue for every product but the projection
    X3Problem p[kMaxProblems];
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 16 fp32 -> three planes of 16 bf16 (8 packed dwords each), exact: x = p1 + p2 + p3
__device__ __forceinline__ void split16(const float (&x)[16], unsigned (&p1)[8], unsigned (&p2)[8], unsigned (&p3)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned h = cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cvt_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        p1[i] = h; p2[i] = m; p3[i] = cvt_pk_bf16(sa, sb);
    }
}

// 16 consecutive k (kb .. kb+15) of one operand row/column `r` (already clamped).  Addresses past K are clamped and the
// values are NOT masked here: a select on a loaded value would make the wave wait for the load where it is issued,
// and these loads are meant to stay in flight across the MFMA phase.  mask16() zeroes k >= K when the chunk is used.
template <bool KMAJOR>
__device__ __forceinline__ void load16(const float* __restrict__ P, int ld, int r, int kb, int K, float (&x)[16]) {
    if (KMAJOR) {                        // K % 4 == 0 (host-checked): a 16-byte group is inside or outside as a whole
        const float* src = P + (int64_t)r * ld;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + min(kb + 4 * q, K - 4));
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = P[(int64_t)min(kb + j, K - 1) * ld + r];
    }
}

__device__ __forceinline__ void mask16(float (&x)[16], int kb, int K) {
    if (kb + 16 <= K) return;
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = kb + j < K ? x[j] : 0.f;
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(X3Batch b) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxProblems; ++i)
        if (i < b.n && (int)blockIdx.x >= b.p[i].wg0) pi = i;
    const X3Problem& p = b.p[pi];
    const int local = blockIdx.x - p.wg0;
    const int ks = local % p.split_k, tile = local / p.split_k;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * kTile, n0 = tn * kTile;
    const int nch = (p.K + kBK - 1) / kBK;
    const int per = (nch + p.split_k - 1) / p.split_k;
    const int c0 = ks * per, c1 = min(nch, c0 + per);

    // staging role: (row, 16-k half) within the 128-row operand block
    const int ra = AK ? (tid >> 1) : (tid & 127), ha = AK ? (tid & 1) : (tid >> 7);
    const int rb = BKM ? (tid >> 1) : (tid & 127), hb = BKM ? (tid & 1) : (tid >> 7);
    const int ga = min(m0 + ra, p.M - 1), gb = min(n0 + rb, p.N - 1);
    unsigned char* wa = smem + ra * kRowBytes + ha * 32;
    unsigned char* wb = smem + (kTile + rb) * kRowBytes + hb * 32;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wr = wave >> 1, wc = wave & 1;
    // fragment read addresses: row (lane & 31) of the wave's 32-row blocks, k-slot 8 * (lane >> 5)
    const unsigned char* fa = smem + (wr * 64 + (lane & 31)) * kRowBytes + (lane >> 5) * 16;
    const unsigned char* fb = smem + (kTile + wc * 64 + (lane & 31)) * kRowBytes + (lane >> 5) * 16;

    // Two chunks of raw operands are in flight at any time (register sets 0 and 1): a chunk's loads are issued two
    // iterations before its split, so their latency is covered by two MFMA phases of this workgroup (and the other
    // workgroup of the CU).  Chunk indices past the slice are clamped: the loop has no branch around a load.
    float xa0[16], xb0[16], xa1[16], xb1[16];
    auto fetch = [&](int c, float (&xa)[16], float (&xb)[16]) {
        const int cc = min(c, c1 - 1);
        load16<AK>(p.A, p.lda, ga, cc * kBK + ha * 16, p.K, xa);
        load16<BKM>(p.B, p.ldb, gb, cc * kBK + hb * 16, p.K, xb);
    };
    auto stage = [&](int c, float (&xa)[16], float (&xb)[16]) {        // split chunk c's raw values into the LDS planes
        unsigned q1[8], q2[8], q3[8];
        mask16(xa, c * kBK + ha * 16, p.K);
        mask16(xb, c * kBK + hb * 16, p.K);
        split16(xa, q1, q2, q3);
        *reinterpret_cast<u32x4*>(wa) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        *reinterpret_cast<u32x4*>(wa + 16) = u32x4{q1[4], q1[5], q1[6], q1[7]};
        *reinterpret_cast<u32x4*>(wa + kPlaneBytes) = u32x4{q2[0], q2[1], q2[2], q2[3]};
        *reinterpret_cast<u32x4*>(wa + kPlaneBytes + 16) = u32x4{q2[4], q2[5], q2[6], q2[7]};
        *reinterpret_cast<u32x4*>(wa + 2 * kPlaneBytes) = u32x4{q3[0], q3[1], q3[2], q3[3]};
        *reinterpret_cast<u32x4*>(wa + 2 * kPlaneBytes + 16) = u32x4{q3[4], q3[5], q3[6], q3[7]};
        split16(xb, q1, q2, q3);
        *reinterpret_cast<u32x4*>(wb) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        *reinterpret_cast<u32x4*>(wb + 16) = u32x4{q1[4], q1[5], q1[6], q1[7]};
        *reinterpret_cast<u32x4*>(wb + kPlaneBytes) = u32x4{q2[0], q2[1], q2[2], q2[3]};
        *reinterpret_cast<u32x4*>(wb + kPlaneBytes + 16) = u32x4{q2[4], q2[5], q2[6], q2[7]};
        *reinterpret_cast<u32x4*>(wb + 2 * kPlaneBytes) = u32x4{q3[0], q3[1], q3[2], q3[3]};
        *reinterpret_cast<u32x4*>(wb + 2 * kPlaneBytes + 16) = u32x4{q3[4], q3[5], q3[6], q3[7]};
    };
    auto products = [&]() {                                              // 2 k-steps x 6 piece products x 2 x 2 tiles
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 am[3][2], bn[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    am[pl][t] = *reinterpret_cast<const bf16x8*>(fa + pl * kPlaneBytes + t * 32 * kRowBytes + s * 32);
                    bn[pl][t] = *reinterpret_cast<const bf16x8*>(fb + pl * kPlaneBytes + t * 32 * kRowBytes + s * 32);
                }
            // smallest terms first; MFMA A operand = M-side piece, B operand = N-side piece: D[m][n], a lane holds
            // one column n and 16 rows m — stores and atomics of a half-wave cover 128 contiguous bytes
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int ia[6] = {1, 2, 0, 1, 0, 0};      // piece of the M-side operand
                constexpr int ib[6] = {1, 0, 2, 0, 1, 0};      // piece of the N-side operand
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ia[term]][i], bn[ib[term]][j], acc[i][j], 0, 0, 0);
            }
        }
    };
    if (c0 < c1) {
        fetch(c0, xa0, xb0);
        fetch(c0 + 1, xa1, xb1);
    }
    for (int c = c0; c < c1; c += 2) {
        stage(c, xa0, xb0);
        __syncthreads();
        fetch(c + 2, xa0, xb0);
        __builtin_amdgcn_sched_barrier(0);
        products();
        __syncthreads();
        if (c + 1 < c1) {
            stage(c + 1, xa1, xb1);
            __syncthreads();
            fetch(c + 3, xa1, xb1);
            __builtin_amdgcn_sched_barrier(0);
            products();
            __syncthreads();
        }
    }

    // epilogue: lane holds column n = (lane & 31) of its 32-column block and rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const bool lead = ks == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + (lane & 31);
        if (n >= p.N) continue;
        float bsum = 0.f;
        if (lead) bsum = (p.bias ? p.bias[n] : 0.f) + (p.bias2 ? p.bias2[n] : 0.f);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float* dst = p.C + (int64_t)m * p.ldc + n;
                const float v = acc[i][j][e] + bsum;
                if (p.split_k > 1) atomicAdd(dst, v);
                else *dst = p.beta ? *dst + v : v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// "Ping-pong" variant: 256 x 128 tile, 8 waves in two groups of four.  Group 0 owns tile rows 0..127, group 1 rows
// 128..255 (wave = 64 x 64).  The LDS holds TWO chunk buffers; in every phase one group runs a chunk's 48 MFMAs per
// wave while the other splits and writes its share of the NEXT chunk (group 1: the 256 A rows, group 0: the 128 B
// rows) — each SIMD hosts one wave of either group, so the matrix pipe and the VALU/LDS-store path of a SIMD are busy
// at the same time by construction, with one barrier per phase and two phases per chunk:
//     phase 2k   : group 0 products(chunk k)      | group 1 stages A(chunk k+1), fetches A(chunk k+3)
//     phase 2k+1 : group 0 stages B(chunk k+1),.. | group 1 products(chunk k)
// Rows are 64 bytes (32 bf16), the 16-byte slots of a row XOR-swizzled (swz() below): the ds_read_b128 lane groups
// (MI355X_MICROARCH.md, LDS) then hit 16 distinct bank quads and the staging stores are conflict-free too.  2 x 3 planes x 384 rows x 64 B = 144 KB: one workgroup
// per CU, raw operands of two chunks in flight per staging thread.
constexpr int kPM = 256, kPN = 128;
constexpr int kPRows = kPM + kPN;
constexpr int kPPlane = kPRows * 64;
constexpr int kPBuf = 3 * kPPlane;                 // 73,728
constexpr int kPLds = 2 * kPBuf;                   // 147,456

// slot ^ f(row), f = ((row >> 2) ^ (row >> 1)) & 3: conflict-free for the fragment reads (every 16-lane group of a
// ds_read_b128 meets four rows of each residue mod 4, and f is distinct on them) AND for both staging patterns'
// ds_write_b128 (8-lane groups = 4 rows x 2 halves, or 8 consecutive rows: same-parity rows get distinct f)
__device__ __forceinline__ int swz(int row, int slot) { return row * 64 + ((slot ^ (((row >> 2) ^ (row >> 1)) & 3)) << 4); }

// BAL ("balanced", accumulating products only: C += A B): the (tile, k-quad) units of ALL tiles of all problems — a
// k-quad = 4 chunks = 128 k — are dealt out evenly over the CUs in tile-major order; a CU's share is cut at tile
// boundaries into at most kBalSegs segments (the tail of a tile, whole tiles, the head of the next), each segment is ONE
// workgroup (blockIdx = segment * shares + share; shares with fewer segments leave empty workgroups that exit at once), and
// every segment adds its partial tile with f32 atomics.  Tiles x chunks need not be a multiple of anything: the decoder's
// weight-gradient group (270 tiles of 40 chunks: two rounds of one tile per CU, the second 5% full) becomes 10-11 quads
// per CU instead of 20.  (The segments of a share as a loop around the pipeline inside one workgroup cost 230 spilled
// registers: hipcc keeps the per-thread roles of both groups alive across the back edge.)
constexpr int kBalSegs = 4;
template <bool AK, bool BKM, bool STATS, bool BAL = false>
__global__ __launch_bounds__(512, 2) void gemm_x3pp_kernel(X3Batch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, t = tid & 255;
    int pi = 0, tile = 0, c0 = 0, c1 = 0;
    bool lead = true;
    if (BAL) {
        // (segment-major: workgroups go to the XCDs round-robin by index — share-major, the first segments of all shares, the
        // only ones most shares have, would all land on two of the eight XCDs)
        const int shares = gridDim.x / kBalSegs, seg = blockIdx.x / shares, share = blockIdx.x - seg * shares;
        int q_cur = (int)((int64_t)share * b.total_q / shares);
        const int q_end = (int)((int64_t)(share + 1) * b.total_q / shares);
        for (int sidx = 0; ; ++sidx) {
            if (q_cur >= q_end) return;                       // this share has fewer segments
            pi = 0;
#pragma unroll
            for (int i = 1; i < kMaxProblems; ++i)
                if (i < b.n && q_cur >= b.p[i].q0) pi = i;
            const int qpt = b.p[pi].qpt, lt = q_cur - b.p[pi].q0;
            tile = lt / qpt;
            const int qq = lt - tile * qpt, take = min(qpt - qq, q_end - q_cur);
            if (sidx == seg) {
                const int nchp = (b.p[pi].K + kBK - 1) / kBK;
                c0 = 4 * qq; c1 = min(nchp, 4 * (qq + take));
                lead = qq == 0;
                break;
            }
            q_cur += take;
        }
    } else {
#pragma unroll
        for (int i = 1; i < kMaxProblems; ++i)
            if (i < b.n && (int)blockIdx.x >= b.p[i].wg0) pi = i;
    }
    const X3Problem& p = b.p[pi];
    const int nch = (p.K + kBK - 1) / kBK;
    if (!BAL) {
        const int local = blockIdx.x - p.wg0;
        const int ks = local % p.split_k;
        tile = local / p.split_k;
        const int per = (nch + p.split_k - 1) / p.split_k;
        c0 = ks * per; c1 = min(nch, c0 + per);
        lead = ks == 0;
    }
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * kPM, n0 = tn * kPN;
    const int n_ch = c1 - c0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int wr = w4 >> 1, wc = w4 & 1;
    const int fr_a = grp * 128 + wr * 64 + (lane & 31), fr_b = kPM + wc * 64 + (lane & 31), fh = lane >> 5;
    const int K = p.K;

    // An item = 16 consecutive k of one operand row.  Loads use a UNIFORM base (scalar registers) plus a per-thread
    // 32-bit byte offset, so a chunk's address arithmetic is two VALU ops per 16-byte load (k-major: clamp to the row's
    // last 16-byte group) or none at all (m-major: the k row is chosen by scalar code).
    //   k-major: off = min(row_off + chunk_bytes + 16 q, row_lim)      m-major: base_j = P + min(k0 + j, K-1) * ld, off = 4 row
    auto fetch_k = [&](const float* P, unsigned row_off, unsigned row_lim, int kb, float (&x)[16]) {
        const unsigned cb = (unsigned)kb * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned off = min(row_off + (cb + 16u * q), row_lim);
            const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(P) + off);
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    };
    // (m-major rows as global loads with a 64-bit scalar row base each.  Tried in round 4: the same loads through a buffer
    // descriptor with the k row as the SCALAR offset — one s_add per load instead of a scalar multiply, a clamp and a 64-bit
    // vector add, 160 instructions fewer per chunk — ran SLOWER: 2.72-2.82 us per 32-k chunk against 2.32-2.55 for the
    // weight-gradient layout, the step 4.03 against 3.95 ms; sixteen loads back to back with nothing between them.)
    auto fetch_m = [&](const float* P, int ld, unsigned col_off, int kb, float (&x)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* rowk = P + (int64_t)min(kb + j, K - 1) * ld;          // uniform
            x[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rowk) + col_off);
        }
    };
    const bool ragged_k = (K % kBK) != 0;
    auto put1 = [&](int c, unsigned char* dst0, unsigned char* dst1, int half, float (&x)[16]) {
#if defined(MMQG_X3_ABLATE) && (MMQG_X3_ABLATE & 2)
        return;
#endif
        const int cg = c0 + min(c, n_ch - 1);
        if (ragged_k && cg == nch - 1) mask16(x, cg * kBK + half * 16, K);      // uniform branch, no loads inside
        unsigned q1[8], q2[8], q3[8];
        split16(x, q1, q2, q3);
        *reinterpret_cast<u32x4*>(dst0) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        *reinterpret_cast<u32x4*>(dst1) = u32x4{q1[4], q1[5], q1[6], q1[7]};
        *reinterpret_cast<u32x4*>(dst0 + kPPlane) = u32x4{q2[0], q2[1], q2[2], q2[3]};
        *reinterpret_cast<u32x4*>(dst1 + kPPlane) = u32x4{q2[4], q2[5], q2[6], q2[7]};
        *reinterpret_cast<u32x4*>(dst0 + 2 * kPPlane) = u32x4{q3[0], q3[1], q3[2], q3[3]};
        *reinterpret_cast<u32x4*>(dst1 + 2 * kPPlane) = u32x4{q3[4], q3[5], q3[6], q3[7]};
    };
    // fragment addresses of this wave within a buffer (the swizzle term is the same for rows r and r + 32)
    const int ra0 = swz(fr_a, fh), ra1 = swz(fr_a, 2 + fh), rb0 = swz(fr_b, fh), rb1 = swz(fr_b, 2 + fh);
    auto products = [&](int buf) {
#if defined(MMQG_X3_ABLATE) && (MMQG_X3_ABLATE & 1)
        return;
#endif
#ifdef MMQG_X3_PRIO          // (compile-time A/B, tools/x3_probe.hip: priority 1 / 3 for the MFMA phase: k-major operands -1..-4 %, m-major +3 %)
        __builtin_amdgcn_s_setprio(MMQG_X3_PRIO);
#endif
        const unsigned char* base = smem + buf * kPBuf;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 am[3][2], bn[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    am[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * kPPlane + (s ? ra1 : ra0) + i * 32 * 64);
                    bn[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * kPPlane + (s ? rb1 : rb0) + i * 32 * 64);
                }
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int ia[6] = {0, 0, 1, 1, 0, 2};      // planes in the order their LDS reads return
                constexpr int ib[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ia[term]][i], bn[ib[term]][j], acc[i][j], 0, 0, 0);
            }
        }
#ifdef MMQG_X3_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // The two groups run SEPARATE loops (same number of barriers): with one loop and a role branch inside, hipcc's
    // wait-count pass sees "a load of the other role may be pending on these registers" at every merge and makes the
    // MFMA phase wait for global loads it never issued.
    if (grp == 0) {
        // stages the B rows: one (row, 16-k half) item per thread
        const int row = BKM ? (t >> 1) : (t & 127), half = BKM ? (t & 1) : (t >> 7);
        const int g = min(n0 + row, p.N - 1);
        const unsigned off = BKM ? ((unsigned)g * (unsigned)p.ldb + half * 16) * 4u : (unsigned)g * 4u;
        const unsigned lim = ((unsigned)g * (unsigned)p.ldb + (unsigned)(K - 4)) * 4u;
        unsigned char* d0[2] = {smem + swz(kPM + row, 2 * half), smem + kPBuf + swz(kPM + row, 2 * half)};
        unsigned char* d1[2] = {smem + swz(kPM + row, 2 * half + 1), smem + kPBuf + swz(kPM + row, 2 * half + 1)};
        const int hk = BKM ? 0 : half * 16;          // m-major: the half picks the k rows (uniform per wave)
        auto fetch = [&](int c, float (&x)[16]) {
#if defined(MMQG_X3_ABLATE) && (MMQG_X3_ABLATE & 4)
            if (c > 2) return;
#endif
            const int kb = (c0 + min(c, n_ch - 1)) * kBK;
            if (BKM) fetch_k(p.B, off, lim, kb, x);
            else fetch_m(p.B, p.ldb, off, kb + hk, x);
        };
        float xe[16], xo[16];
        if (n_ch > 0) { fetch(0, xe); fetch(1, xo); put1(0, d0[0], d1[0], half, xe); fetch(2, xe); }
        __syncthreads();
        for (int k = 0; k < n_ch; k += 2) {
            products(0);
            __syncthreads();
            put1(k + 1, d0[1], d1[1], half, xo); fetch(k + 3, xo);
            __syncthreads();
            if (k + 1 >= n_ch) break;
            products(1);
            __syncthreads();
            put1(k + 2, d0[0], d1[0], half, xe); fetch(k + 4, xe);
            __syncthreads();
        }
    } else {
        // stages the A rows: two items per thread
        const int row_a = AK ? (t >> 1) : t, row_b = AK ? 128 + (t >> 1) : t;
        const int half_a = AK ? (t & 1) : 0, half_b = AK ? (t & 1) : 1;
        const int g_a = min(m0 + row_a, p.M - 1), g_b = min(m0 + row_b, p.M - 1);
        const unsigned off_a = AK ? ((unsigned)g_a * (unsigned)p.lda + half_a * 16) * 4u : (unsigned)g_a * 4u;
        const unsigned off_b = AK ? ((unsigned)g_b * (unsigned)p.lda + half_b * 16) * 4u : (unsigned)g_b * 4u;
        const unsigned lim_a = ((unsigned)g_a * (unsigned)p.lda + (unsigned)(K - 4)) * 4u;
        const unsigned lim_b = ((unsigned)g_b * (unsigned)p.lda + (unsigned)(K - 4)) * 4u;
        unsigned char* da0[2] = {smem + swz(row_a, 2 * half_a), smem + kPBuf + swz(row_a, 2 * half_a)};
        unsigned char* da1[2] = {smem + swz(row_a, 2 * half_a + 1), smem + kPBuf + swz(row_a, 2 * half_a + 1)};
        unsigned char* db0[2] = {smem + swz(row_b, 2 * half_b), smem + kPBuf + swz(row_b, 2 * half_b)};
        unsigned char* db1[2] = {smem + swz(row_b, 2 * half_b + 1), smem + kPBuf + swz(row_b, 2 * half_b + 1)};
        auto fetch = [&](int c, float (&x0)[16], float (&x1)[16]) {
#if defined(MMQG_X3_ABLATE) && (MMQG_X3_ABLATE & 4)
            if (c > 2) return;
#endif
            const int kb = (c0 + min(c, n_ch - 1)) * kBK;
            if (AK) { fetch_k(p.A, off_a, lim_a, kb, x0); fetch_k(p.A, off_b, lim_b, kb, x1); }
            else { fetch_m(p.A, p.lda, off_a, kb, x0); fetch_m(p.A, p.lda, off_b, kb + 16, x1); }
        };
        // m-major A with a column-sum request (first column tile only): this thread owns row m0 + t for both k halves,
        // so the sum over k of its raw values is the row's sum — taken on the staged registers, after the tail mask
        const bool want_sum = !AK && p.colsum != nullptr && tn == 0;
        float rsum = 0.f;
        auto tally = [&](int c, const float (&x0)[16], const float (&x1)[16]) {
            if (!want_sum || c >= n_ch) return;                         // uniform
#pragma unroll
            for (int j = 0; j < 16; ++j) rsum += x0[j] + x1[j];
        };
        float xe0[16], xe1[16], xo0[16], xo1[16];
        if (n_ch > 0) {
            fetch(0, xe0, xe1); fetch(1, xo0, xo1);
            put1(0, da0[0], da1[0], half_a, xe0); put1(0, db0[0], db1[0], half_b, xe1); tally(0, xe0, xe1);
            fetch(2, xe0, xe1);
        }
        __syncthreads();
        for (int k = 0; k < n_ch; k += 2) {
            put1(k + 1, da0[1], da1[1], half_a, xo0); put1(k + 1, db0[1], db1[1], half_b, xo1); tally(k + 1, xo0, xo1);
            fetch(k + 3, xo0, xo1);
            __syncthreads();
            products(0);
            __syncthreads();
            if (k + 1 >= n_ch) break;
            put1(k + 2, da0[0], da1[0], half_a, xe0); put1(k + 2, db0[0], db1[0], half_b, xe1); tally(k + 2, xe0, xe1);
            fetch(k + 4, xe0, xe1);
            __syncthreads();
            products(1);
            __syncthreads();
        }
        if (want_sum && m0 + t < p.M) {
            atomicAdd(p.colsum + m0 + t, rsum);
            if (p.colsum2) atomicAdd(p.colsum2 + m0 + t, rsum);
        }
    }

    // Epilogue through the LDS (free now): the tile comes back row by row, a half-wave per row, 16 bytes per lane — 512
    // contiguous bytes per row and instruction.  The projection (STATS: decoder.py:106 + train.py:174; one k slice, no beta)
    // also takes bias, the row's max / first argmax / sum-exp over the tile's 128 columns there.  Every other product takes
    // its C += / = / atomic-add in that shape too when C allows 16-byte accesses: with a lane holding one column and 16
    // rows the accumulating epilogue was 64 dependent dword load-add-store rounds (33 us of a 256 x 128 tile's time,
    // against 14 for plain dword stores: tools/x3_fixed_cost.py).
    const bool c_vec = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && !b.dword_epilogue;
    if (STATS || c_vec) {
        constexpr int kTS = 132;                             // floats per staged row (128 + 4: ds_read_b128-aligned)
        float* T = reinterpret_cast<float*>(smem);
        __syncthreads();                                     // every wave is done reading the operand buffers
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = grp * 128 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    T[r * kTS + wc * 64 + j * 32 + (lane & 31)] = acc[i][j][e];
                }
        __syncthreads();
        // a half-wave owns a row at a time: lane l of it holds columns 4 l .. 4 l + 3 (512 contiguous bytes per row and
        // store), the row's statistics are reduced over the 32 lanes
        const int l32 = tid & 31, hw = tid >> 5;
        const int n = n0 + 4 * l32;
        f32x4 bias4{0.f, 0.f, 0.f, 0.f};
        if (STATS || lead) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.N) bias4[e] = (p.bias ? p.bias[n + e] : 0.f) + (p.bias2 ? p.bias2[n + e] : 0.f);
        }
        const bool vec_ok = c_vec && n + 3 < p.N;
        const bool atomic = !STATS && (BAL || p.split_k > 1), accum = !STATS && p.beta != 0;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = hw + 16 * it;
            const int m = m0 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(T + row * kTS + 4 * l32) + bias4;
            if (!STATS) {
                if (m >= p.M) continue;
                float* crow = p.C + (int64_t)m * p.ldc;
                if (atomic) {
                    // (f32 atomics run at full rate for 128 contiguous bytes per half-wave, at a quarter of it for 4 of
                    // every 16 bytes: here lane l takes columns l, l + 32, l + 64, l + 96 of the row)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int nn = n0 + 32 * e + l32;
                        if (nn < p.N) {
                            float x = T[row * kTS + 32 * e + l32];
                            if (lead) x += (p.bias ? p.bias[nn] : 0.f) + (p.bias2 ? p.bias2[nn] : 0.f);
                            atomicAdd(crow + nn, x);
                        }
                    }
                } else if (vec_ok) {
                    if (accum) v += *reinterpret_cast<const f32x4*>(crow + n);
                    *reinterpret_cast<f32x4*>(crow + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) crow[n + e] = accum ? crow[n + e] + v[e] : v[e];
                }
                continue;
            }
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= p.N) v[e] = -INFINITY;          // past N: out of the statistics, never stored
                if (v[e] > best) { best = v[e]; bi = n + e; }            // ascending n: the first maximum stays
            }
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                const float ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            float sum = expf(v[0] - best) + expf(v[1] - best) + expf(v[2] - best) + expf(v[3] - best);
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if (m < p.M) {
                if (l32 == 0) p.stats[(int64_t)m * p.tiles_n + tn] = make_float4(best, sum, __int_as_float(bi), 0.f);
                float* crow = p.C + (int64_t)m * p.ldc;
                if (vec_ok) *reinterpret_cast<f32x4*>(crow + n) = v;
                else
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) crow[n + e] = v[e];
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + (lane & 31);
        if (n >= p.N) continue;
        float bsum = 0.f;
        if (lead) bsum = (p.bias ? p.bias[n] : 0.f) + (p.bias2 ? p.bias2[n] : 0.f);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + grp * 128 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float* dst = p.C + (int64_t)m * p.ldc + n;
                const float v = acc[i][j][e] + bsum;
                if (BAL || p.split_k > 1) atomicAdd(dst, v);
                else *dst = p.beta ? *dst + v : v;
            }
    }
}

__global__ void x3_zero_kernel(float* C, int ldc, int M, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    C[(i / N) * ldc + (i % N)] = 0.f;
}

int x3_cus() {
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        return pr.multiProcessorCount;
    }();
    return n;
}

bool aligned16p(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <bool AK, bool BKM, bool STATS = false, bool BAL = false>
int launch_pp(const X3Batch& b, int wgs, hipStream_t s) {
    static int attr = 0;
    if (attr == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3pp_kernel<AK, BKM, STATS, BAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kPLds);
        if (e != hipSuccess) (void)hipGetLastError();
        attr = e == hipSuccess ? 1 : -1;
    }
    if (attr < 0) return 1;
    hipLaunchKernelGGL((gemm_x3pp_kernel<AK, BKM, STATS, BAL>), dim3(wgs), dim3(512), (size_t)kPLds, s, b);
    return check_launch("gemm_x3pp");
}

}  // namespace

namespace mmqg {

// workgroup slots of one round of the ping-pong kernel (one 256 x 128 tile per CU); 0 = the kernel is not available
int gemm_x3_slots() {
    static const bool pp = [] { const char* e = getenv("MMQG_X3_PP"); return !e || atoi(e) != 0; }();
    return pp ? x3_cus() : 0;
}

bool gemm_x3_enabled() {
    static const bool on = [] { const char* e = getenv("MMQG_GEMM_X3"); return !e || atoi(e) != 0; }();
    return on;
}

// worth taking: big enough that the tile grid (with k slices) fills the chip and the arithmetic dominates; short-K
// products (the embedding-width hoists, K = 300) run as fast on the fp32 MFMA kernels (tools/x3_check.py)
bool gemm_x3_wants(int M, int N, int K) {
    // MMQG_X3_MIN_K: shortest K the split-bf16 kernel takes (A/B switch; the hoisted K = 300 products of the step)
    static const int min_k = [] { const char* e = getenv("MMQG_X3_MIN_K"); return e ? atoi(e) : 512; }();
    return gemm_x3_enabled() && M >= 96 && N >= 96 && K >= min_k && (double)M * N * K >= 2.5e8;
}

// Vocabulary projection logits = h W^T + bias with the loss's row statistics per 128-column tile (same layout as
// gemm_nt_tile's: [M][ceil(N/128)] float4).  0 = launched (*stats_tiles set), 1 = not taken.
int gemm_x3_projection(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias, float* C,
                       int ldc, float* stats, int64_t stats_bytes, int* stats_tiles, hipStream_t s) {
    if (stats_tiles) *stats_tiles = 0;
    static const bool pp = [] { const char* e = getenv("MMQG_X3_PP"); return !e || atoi(e) != 0; }();
    if (!pp || !stats || !stats_tiles || !gemm_x3_wants(M, N, K) || x3_cus() < 64) return 1;
    if (!((lda % 4 == 0) && (ldb % 4 == 0) && (K % 4 == 0) && aligned16p(A) && aligned16p(B) && aligned16p(stats))) return 1;
    if ((int64_t)M * lda >= (1ll << 29) || (int64_t)N * ldb >= (1ll << 29)) return 1;
    const int tiles_n = ceil_div(N, kPN);
    if (stats_bytes < (int64_t)M * tiles_n * 16) return 1;
    X3Batch b{};
    b.n = 1;
    X3Problem& p = b.p[0];
    p.M = M; p.N = N; p.K = K; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.bias = bias; p.bias2 = nullptr; p.beta = 0; p.split_k = 1; p.tiles_n = tiles_n; p.wg0 = 0;
    p.stats = reinterpret_cast<float4*>(stats);
    const int rc = launch_pp<true, true, true>(b, ceil_div(M, kPM) * tiles_n, s);
    if (rc == 0) *stats_tiles = tiles_n;
    return rc;
}

// 0 = launched; 1 = not taken (caller falls back); < 0 error.  All problems share one layout pair.
int gemm_x3_grouped(int a_layout, int b_layout, const GemmProblem* probs, const float* const* bias, const float* const* bias2,
                    int n, hipStream_t s, float* const* colsum, float* const* colsum2) {
    if (n <= 0) return 0;
    if (!gemm_x3_enabled() || x3_cus() < 64) return 1;
    if (a_layout == MMQG_MN_MAJOR && b_layout == MMQG_K_MAJOR) return 1;       // no caller; not instantiated
    static const bool pp = [] { const char* e = getenv("MMQG_X3_PP"); return !e || atoi(e) != 0; }();
    if (colsum && !(pp && a_layout == MMQG_MN_MAJOR)) return 1;                // fused column sums: ping-pong kernel, m-major A
    const int tile_m = pp ? kPM : kTile, tile_n = pp ? kPN : kTile;
    // every problem is checked BEFORE anything is launched: "not taken" must leave all outputs untouched
    for (int i = 0; i < n; ++i) {
        const GemmProblem& q = probs[i];
        if (q.M <= 0 || q.N <= 0 || q.K < 4) return 1;
        // per-thread byte offsets are 32-bit; k-major operands are read by 16-byte loads
        if ((int64_t)q.M * q.K >= (1ll << 29) || (int64_t)q.N * q.K >= (1ll << 29)) return 1;
        if ((a_layout == MMQG_K_MAJOR ? (int64_t)q.M * q.lda : (int64_t)q.K * q.lda) >= (1ll << 29)) return 1;
        if ((b_layout == MMQG_K_MAJOR ? (int64_t)q.N * q.ldb : (int64_t)q.K * q.ldb) >= (1ll << 29)) return 1;
        if (a_layout == MMQG_K_MAJOR && !((q.lda % 4 == 0) && (q.K % 4 == 0) && aligned16p(q.A))) return 1;
        if (b_layout == MMQG_K_MAJOR && !((q.ldb % 4 == 0) && (q.K % 4 == 0) && aligned16p(q.B))) return 1;
    }
    for (int g0 = 0; g0 < n; g0 += kMaxProblems) {
        const int ng = std::min(kMaxProblems, n - g0);
        static const int dword_epi = [] { const char* e = getenv("MMQG_X3_DWORD_EPILOGUE"); return e && atoi(e) != 0 ? 1 : 0; }();
        X3Batch b{};
        b.n = ng;
        b.dword_epilogue = dword_epi;
        int64_t tiles_total = 0;
        for (int i = 0; i < ng; ++i) {
            const GemmProblem& q = probs[g0 + i];
            tiles_total += (int64_t)ceil_div(q.M, tile_m) * ceil_div(q.N, tile_n);
        }
        // Accumulating weight-gradient groups (both operands m/n-major, every C += ...): one workgroup per CU and the k-quads
        // of all tiles dealt out evenly (BAL above) — no tile-count quantisation, no read-modify-write epilogue.
        // OPT-IN (MMQG_X3_BAL=1).  Measured (round 4): alone, a half-filled launch gains (2048 x 2048 x 2048, 128 tiles: 176 -> 137 us),
        // but in the step every weight-gradient group loses (text encoder 728 -> 820 us, frame encoder 140 -> 171 us per phase):
        // two atomic epilogues per CU instead of one 16-byte read-modify-write, and workgroups of unequal length dealt out greedily.
        static const bool bal_on = [] { const char* e = getenv("MMQG_X3_BAL"); return e && atoi(e) != 0; }();
        bool all_beta = pp && bal_on && a_layout == MMQG_MN_MAJOR && b_layout == MMQG_MN_MAJOR;
        for (int i = 0; i < ng && all_beta; ++i) all_beta = probs[g0 + i].beta != 0;
        if (all_beta) {
            int q = 0;
            for (int i = 0; i < ng; ++i) {
                const GemmProblem& gp = probs[g0 + i];
                X3Problem& p = b.p[i];
                p.M = gp.M; p.N = gp.N; p.K = gp.K;
                p.A = gp.A; p.lda = gp.lda; p.B = gp.B; p.ldb = gp.ldb; p.C = gp.C; p.ldc = gp.ldc;
                p.bias = bias ? bias[g0 + i] : nullptr; p.bias2 = bias2 ? bias2[g0 + i] : nullptr;       // (added by a tile's first segment)
                p.colsum = colsum ? colsum[g0 + i] : nullptr; p.colsum2 = (colsum && colsum2) ? colsum2[g0 + i] : nullptr;
                p.beta = 1; p.split_k = 1; p.wg0 = 0;
                p.tiles_n = ceil_div(gp.N, tile_n);
                p.qpt = ceil_div(ceil_div(gp.K, kBK), 4);
                p.q0 = q;
                q += ceil_div(gp.M, tile_m) * p.tiles_n * p.qpt;
            }
            b.total_q = q;
            const int shares = std::min(x3_cus(), q);
            // a share of R quads crosses at most 2 + R / (quads per tile) tile boundaries
            int min_qpt = 1 << 30;
            for (int i = 0; i < ng; ++i) min_qpt = std::min(min_qpt, b.p[i].qpt);
            const bool fits = ceil_div(q, shares) / min_qpt + 2 <= kBalSegs;
            const int rc = fits ? launch_pp<false, false, false, true>(b, kBalSegs * shares, s) : 1;
            if (fits) {
                if (rc > 0 && g0 > 0) { set_error("gemm_x3: launch refused after part of the group ran"); return -1; }
                if (rc != 0) return rc;
                continue;
            }
            b = X3Batch{};                     // (a share would span too many tiles: one tile per workgroup below)
            b.n = ng;
            b.dword_epilogue = dword_epi;
        }
        // k slices: fill the chip's workgroup slots once, keep >= 8 chunks per slice
        const int slots = (pp ? 1 : 2) * x3_cus();
        int wg = 0;
        for (int i = 0; i < ng; ++i) {
            const GemmProblem& q = probs[g0 + i];
            X3Problem& p = b.p[i];
            p.M = q.M; p.N = q.N; p.K = q.K;
            p.A = q.A; p.lda = q.lda; p.B = q.B; p.ldb = q.ldb; p.C = q.C; p.ldc = q.ldc;
            p.bias = bias ? bias[g0 + i] : nullptr; p.bias2 = bias2 ? bias2[g0 + i] : nullptr;
            p.colsum = colsum ? colsum[g0 + i] : nullptr; p.colsum2 = (colsum && colsum2) ? colsum2[g0 + i] : nullptr;
            p.beta = q.beta ? 1 : 0;
            p.tiles_n = ceil_div(q.N, tile_n);
            const int tiles = ceil_div(q.M, tile_m) * p.tiles_n;
            const int nch = ceil_div(q.K, kBK);
            int split = 1;
            if (tiles_total < slots) {
                // as many slices as fit ONE round of workgroups (a partly filled second round costs a whole one), at most
                // 4: every slice adds its 256 x 128 partial tile with f32 atomics.  Alone, a thin product likes more (tools/
                // x3_split.py: the 1280 x 512 x 10000 data gradient takes 127 us at 8 slices, 143 at 13, 205 at 4); in the
                // step, where a second queue has a product of its own for the CUs a launch leaves free, fewer and longer
                // slices win: 3.870 / 3.883 / 3.895 ms per step at 4 against 3.907 / 3.911 / 3.899 at 8 (3.95 at 3, 4.11 at 2;
                // MMQG_X3_MAX_SPLIT overrides)
                static const int max_split = [] { const char* e = getenv("MMQG_X3_MAX_SPLIT"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
                split = (int)std::min<int64_t>(std::max<int64_t>(slots / tiles_total, 1), max_split);
                split = std::min(split, std::max(1, nch / 8));
            }
            p.split_k = split;
            p.wg0 = wg;
            wg += tiles * split;
            if (split > 1 && !p.beta) {
                const int64_t tot = (int64_t)q.M * q.N;
                hipLaunchKernelGGL(x3_zero_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, q.C, q.ldc, q.M, q.N);
                p.beta = 1;
            }
        }
        if (pp) {
            int rc;
            if (a_layout == MMQG_K_MAJOR && b_layout == MMQG_K_MAJOR) rc = launch_pp<true, true>(b, wg, s);
            else if (a_layout == MMQG_K_MAJOR) rc = launch_pp<true, false>(b, wg, s);
            else rc = launch_pp<false, false>(b, wg, s);
            if (rc > 0 && g0 > 0) { set_error("gemm_x3: launch refused after part of the group ran"); return -1; }
            if (rc != 0) return rc;
            continue;
        }
        if (a_layout == MMQG_K_MAJOR && b_layout == MMQG_K_MAJOR)
            hipLaunchKernelGGL((gemm_x3_kernel<true, true>), dim3(wg), dim3(256), 0, s, b);
        else if (a_layout == MMQG_K_MAJOR)
            hipLaunchKernelGGL((gemm_x3_kernel<true, false>), dim3(wg), dim3(256), 0, s, b);
        else
            hipLaunchKernelGGL((gemm_x3_kernel<false, false>), dim3(wg), dim3(256), 0, s, b);
        MMQG_TRY(check_launch("gemm_x3"));
    }
    return 0;
}

}  // namespace mmqg
