// Persistent forward time loop of a stacked LSTM (TextEncoder, model/encoder.py:80-111 driven by
// train.py:159-166; the LSTM stage of VideoConvLstmEncoder, encoder.py:54,69): ONE launch runs all
// T + L - 1 wavefront diagonals, the recurrent weights stay in LDS for the whole sequence and the
// workgroups meet at a device-wide barrier once per diagonal (grid_barrier.h, 2.3 us) instead of at a
// kernel boundary (~5 us of drain + argument fetch + cold operand fetch per launch, profiles/
// r01_skinny_stage_probe.txt).
//
// Work split (built on the host, passed by value): layer l is cut into H/4 "units" of 4 hidden units =
// 16 gate columns (i,f,g,o of each) with K_l = H (layer 0: its input product is hoisted into one GEMM
// over all T*B rows) or 2H (layers >= 1: [x | h] against [W_ih | W_hh]); a unit's batch rows come in
// blocks of 16.  Every workgroup (one per CU) owns up to 2 (unit, row-block range) tasks of about equal
// total cost and keeps their weight rows in LDS in MFMA-fragment order; at config 2 (L=3, H=512, B=64)
// that is one whole unit of layer 1 or 2 (64 KB) plus half the rows of a layer-0 unit (32 KB) per CU: the
// 20 MB of recurrent weights are read from memory once per sequence instead of once per step.
//
// Per diagonal a task computes   pre[16 gate columns][rows] = W_frag * [x(t) | h(t-1)]^T
// with v_mfma_f32_16x16x4_f32 (A operand = weights, so one lane ends up holding the four gates of one
// (row, hidden unit): the cell update needs no exchange), keeps c and h of its rows in registers, and
// publishes h(t) (and the dropped h for the layer above) in an exchange buffer laid out [k/4][64 rows][4]
// so that consumers fetch MFMA operand fragments with whole-line 16-byte loads.  Exchange stores / loads
// are sc1 (write-through / L1-bypassing), which is what lets the barrier go without fences.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "grid_barrier.h"
#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

using namespace mmqg;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxTasks = 2;
constexpr int kMaxL = 3;               // layers a persistent stack may have (the 2-bit layer field of a task)
constexpr int kMaxWG = 256;
constexpr int kThreads = 512;          // 8 waves: 2 per SIMD
constexpr int kWaves = kThreads / 64;
constexpr int kRows = 64;              // rows of the exchange layout (B <= 64)
constexpr int kLdsBudget = 160 * 1024 - 1024;   // dynamic LDS a workgroup may use (the kernel has a little static LDS too)

// layer | unit << 2 | first row block << 14 | row blocks << 17 (0 row blocks = no task)
struct Plan { uint32_t task[kMaxWG * kMaxTasks]; };

struct PersistArgs {
    int T, B, L, H, G;
    const float* w_ih[kMaxL]; const float* w_hh[kMaxL];
    const float* b_ih[kMaxL]; const float* b_hh[kMaxL];
    const float* h0; const float* c0; const int32_t* lens;
    float* gates; float* hs; float* cs; float* hdrop; float* y; int64_t y_stride_t, y_stride_b;
    float drop_p; int drop; uint64_t seed, stream_base; const int32_t* seed_off;
    float* hx;            // [L][2][H/4][64][4]   h(t) of every layer, slot = t & 1
    float* xd;            // [L-1][2][H/4][64][4] dropped h(t) for the layer above (only when dropout is live)
    gb::XBar* bar;
    float* poison;        // receives NaN when a barrier times out
    unsigned* sticky_fail;   // device counter of failed launches (never reset by the library), in the workspace
    unsigned* host_fail;     // pinned host word the library polls without synchronising (nullable)
    unsigned expect_wg, max_spins;   // workgroups the barrier waits for (= gridDim.x; the test hook adds to it) / spin bound
    unsigned long long* trace;   // diagnostics (tools/persist_trace.py): [workgroup][diagonal][4] wall-clock stamps, or null
};

__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const int32_t* off) {
    return off ? seed + (uint64_t)(uint32_t)off[0] * 0x9E3779B97F4A7C15ull : seed;
}

// select by a run-time (uniform) layer index without indexing the by-value argument array at run time
// (that makes hipcc keep a scratch copy of the array)
template <typename P>
__device__ __forceinline__ P pick(P const (&arr)[kMaxL], int l) {
    static_assert(kMaxL == 3, "pick() lists three layers");
    return l == 0 ? arr[0] : (l == 1 ? arr[1] : arr[2]);
}

struct TaskInfo { int layer, unit, mb0, nmb, K, nch, msplit, ksplit, woff; };

__device__ __forceinline__ TaskInfo decode_task(uint32_t w, int H) {
    TaskInfo t;
    t.layer = w & 3; t.unit = (w >> 2) & 0xFFF; t.mb0 = (w >> 14) & 7; t.nmb = (w >> 17) & 7;
    t.K = t.layer == 0 ? H : 2 * H;
    t.nch = t.nmb ? t.K / 16 : 0;
    t.msplit = t.nmb >= 3 ? 4 : (t.nmb == 2 ? 2 : 1);
    t.ksplit = kWaves / t.msplit;
    t.woff = 0;
    return t;
}

#define MMQG_PSTAMP(slot)                                                                              \
    if (TRACE && tid == 0) a.trace[((size_t)blockIdx.x * (T + L - 1) + s) * 6 + (slot)] = wall_clock64();

constexpr int kRing0 = 16, kRing1 = 4;     // operand chunks (1 KB each) in flight per wave: first task / second task

// One wave's share of one task on one diagonal: `n` k-chunks (16 k each) starting at chunk c_lo, for one block
// of 16 rows.  The chunk count is padded to a multiple of the ring depth with chunks whose weight fragment is the
// all-zero chunk at `zoff`, so the loops below contain no branch at all: hipcc then counts the outstanding
// loads exactly (s_waitcnt vmcnt(7) in the steady state); with any conditional inside it falls back to
// vmcnt(0) before every use and the loads and the MFMAs stop overlapping.
struct WaveTask {
    int n, n_pad, c_lo, hc, off_x, off_h, lane_off, woff;
};

template <typename Rsrc>
__device__ __forceinline__ f32x4 load_chunk(const Rsrc& rs, const WaveTask& w, int i) {
    const int c = w.c_lo + min(i, w.n - 1);
    const int off = (c < w.hc ? w.off_x + c * (4 * kRows * 16) : w.off_h + (c - w.hc) * (4 * kRows * 16)) + w.lane_off;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
}

// weight fragment of the wave's i-th chunk (the zero chunk at LDS offset 0 for padding chunks)
__device__ __forceinline__ f32x4 weight_frag(const f32x4* lds, const WaveTask& w, int lane, int i) {
    return lds[(i < w.n ? w.woff + (w.c_lo + i) * 64 : 0) + lane];
}

// 16 k of one chunk: two independent accumulator chains (the 16x16x4 f32 MFMA has 40 cycles of dependent latency
// against 32 of issue; one chain per wave leaves the matrix pipe idle whenever the SIMD's other wave is waiting)
__device__ __forceinline__ void mfma_chunk(f32x4& acc0, f32x4& acc1, const f32x4& wt, const f32x4& x) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.x, x.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.y, x.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.z, x.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt.w, x.w, acc1, 0, 0, 0);
}

template <int R, typename Rsrc>
__device__ __forceinline__ void ring_fill(f32x4 (&ring)[R], const Rsrc& rs, const WaveTask& w) {
#pragma unroll
    for (int d = 0; d < R; ++d) ring[d] = load_chunk(rs, w, d);
}

// all chunks of one wave's share of a task: R chunks of operand loads stay in flight (w.n_pad is a multiple of R)
template <int R, typename Rsrc>
__device__ __forceinline__ f32x4 ring_products(f32x4 (&ring)[R], const Rsrc& rs, const WaveTask& w, const f32x4* lds, int lane) {
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    // the weight fragment of chunk i + 1 is read from LDS before the MFMAs of chunk i
    f32x4 wcur = weight_frag(lds, w, lane, 0);
    for (int i0 = 0; i0 + R < w.n_pad; i0 += R) {
#pragma unroll
        for (int d = 0; d < R; ++d) {
            const f32x4 wnext = weight_frag(lds, w, lane, i0 + d + 1);
            mfma_chunk(acc0, acc1, wcur, ring[d]);
            ring[d] = load_chunk(rs, w, i0 + R + d);
            wcur = wnext;
        }
    }
#pragma unroll
    for (int d = 0; d < R; ++d) {
        const f32x4 wnext = weight_frag(lds, w, lane, w.n_pad - R + min(d + 1, R - 1));
        mfma_chunk(acc0, acc1, wcur, ring[d]);
        wcur = wnext;
    }
    acc0.x += acc1.x; acc0.y += acc1.y; acc0.z += acc1.z; acc0.w += acc1.w;
    return acc0;
}

template <bool TRACE>
__global__ __launch_bounds__(kThreads, 2) void lstm_persist_fwd_kernel(PersistArgs a, Plan plan) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: everything derived from it stays scalar
    const int T = a.T, B = a.B, L = a.L, H = a.H;
    const int slot_f = kRows * H;                 // floats per exchange slot

    TaskInfo tk[kMaxTasks];
    int wtotal = 64;                              // LDS float4 [0, 64): the all-zero weight chunk
#pragma unroll
    for (int ti = 0; ti < kMaxTasks; ++ti) {
        tk[ti] = decode_task(plan.task[blockIdx.x * kMaxTasks + ti], H);
        tk[ti].woff = wtotal;
        wtotal += tk[ti].nch * 64;
    }
    f32x4* scratch = lds + wtotal;                // [task][wave][lane] partial tiles of the k-split waves
    if (tid < 64) lds[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    // exchange buffers: hx [L][2][slot] directly followed by xd [L-1][2][slot], one descriptor over both
    // (sc1 loads / stores carry aux = 16)
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.hx, 0, (2 * L - 1) * 2 * slot_f * 4, 0x00020000);
    const int xd_base = L * 2 * slot_f * 4;       // byte offset of xd inside the descriptor

    // ---- per-task roles of this wave and the state its cell lanes keep in registers for the whole sequence
    const int j = lane & 15, q = lane >> 4;       // lane (row j of the block, k-group / hidden unit q)
    int mbi[kMaxTasks], ks[kMaxTasks], row[kMaxTasks], len[kMaxTasks];
    bool cellw[kMaxTasks];                        // this wave does the cell update of its row block of the task
    float hreg[kMaxTasks], creg[kMaxTasks], bias[kMaxTasks][4];
    WaveTask wt[kMaxTasks];
#pragma unroll
    for (int ti = 0; ti < kMaxTasks; ++ti) {
        const TaskInfo& t = tk[ti];
        mbi[ti] = wave % t.msplit; ks[ti] = wave / t.msplit;
        row[ti] = (t.mb0 + mbi[ti]) * 16 + j;
        hreg[ti] = 0.f; creg[ti] = 0.f; len[ti] = T;
        bias[ti][0] = bias[ti][1] = bias[ti][2] = bias[ti][3] = 0.f;
        // this wave's k-chunk range of the task (none when it has no row block of it)
        const int per = (t.nch + t.ksplit - 1) / t.ksplit;
        WaveTask& w = wt[ti];
        w.c_lo = min(t.nch, ks[ti] * per);
        w.n = (t.nmb && mbi[ti] < t.nmb) ? min(t.nch, w.c_lo + per) - w.c_lo : 0;
        const int ring = ti == 0 ? kRing0 : kRing1;
        w.n_pad = (w.n + ring - 1) / ring * ring;
        w.hc = t.layer == 0 ? 0 : H / 16;                                 // chunks of the x part
        w.lane_off = ((q * kRows) + (t.mb0 + mbi[ti]) * 16 + j) * 16;     // group (4c + q), row (mb*16 + j)
        w.woff = t.woff; w.off_x = 0; w.off_h = 0;
        cellw[ti] = t.nmb && mbi[ti] < t.nmb && ks[ti] == (ti == 0 ? 0 : t.ksplit - 1);
        if (!cellw[ti]) continue;
        const int u = 4 * t.unit + q, b = row[ti];
        if (b < B) {
            if (a.h0) hreg[ti] = a.h0[((int64_t)t.layer * B + b) * H + u];
            if (a.c0) creg[ti] = a.c0[((int64_t)t.layer * B + b) * H + u];
            if (a.lens) len[ti] = a.lens[b];
        }
        if (t.layer > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[ti][r] = pick(a.b_ih, t.layer)[r * H + u] + pick(a.b_hh, t.layer)[r * H + u];
        }
        // h(-1) into slot 1 of the layer's exchange buffer
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hreg[ti]), rs,
                                              (((t.layer * 2 + 1) * slot_f) + (t.unit * kRows + b) * 4 + q) * 4, 0, 16);
    }

    // h(-1) is on its way: once every wave's stores are acknowledged the launch announces itself at the flat start-up barrier,
    // which turns while the weights are read — when it has turned, every workgroup's h(-1) is published, so no second
    // barrier is needed before the first diagonal
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gb::Ctx bar;
    gb::init_arrive(bar, a.bar, a.max_spins);

    // ---- weights -> LDS, fragment order: chunk c (16 k), lane l = (i = l & 15: gate column (unit i>>2, gate i&3),
    // kq = l >> 4): the 4 consecutive k = 16c + 4kq + {0..3} of that weight row
#pragma unroll
    for (int ti = 0; ti < kMaxTasks; ++ti) {
        const TaskInfo& t = tk[ti];
        const int n = t.nch * 64;
        for (int idx = tid; idx < n; idx += kThreads) {
            const int c = idx >> 6, l = idx & 63, i = l & 15, kq = l >> 4;
            const int row = (i & 3) * H + 4 * t.unit + (i >> 2);
            const int k = 16 * c + 4 * kq;
            const float* src;
            if (t.layer == 0) src = a.w_hh[0] + (int64_t)row * H + k;
            else if (k < H) src = pick(a.w_ih, t.layer) + (int64_t)row * H + k;
            else src = pick(a.w_hh, t.layer) + (int64_t)row * H + (k - H);
            lds[t.woff + idx] = *reinterpret_cast<const f32x4*>(src);
        }
    }

    bool ok = gb::init_wait(bar, a.expect_wg);        // (its workgroup barrier also closes the weight fill)
    // the second-dispatched half of an 8-wave workgroup loses the issue arbitration on every SIMD to its older partner
    // (MI355X_MICROARCH.md, two waves per SIMD, item 4): one static priority for waves 4-7 evens the two halves out
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    const uint64_t seed = a.drop ? eff_seed(a.seed, a.seed_off) : 0;
    for (int s = 0; ok && s < T + L - 1; ++s) {
        MMQG_PSTAMP(0)
        f32x4 acc[kMaxTasks];
        float pre_g[kMaxTasks][4];
        bool on[kMaxTasks];
        f32x4 ring0[kRing0], ring1[kRing1];
        static_assert(kMaxTasks == 2, "the product phase below spells out two tasks");
        // ---- the first ring of every task goes out before any arithmetic
#pragma unroll
        for (int ti = 0; ti < kMaxTasks; ++ti) {
            const TaskInfo& t = tk[ti];
            acc[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
            pre_g[ti][0] = pre_g[ti][1] = pre_g[ti][2] = pre_g[ti][3] = 0.f;
            const int tt = s - t.layer;
            on[ti] = t.nmb && tt >= 0 && tt < T && mbi[ti] < t.nmb;
            if (!on[ti]) continue;
            WaveTask& w = wt[ti];
            // x part: the layer below at time tt (its dropped copy when dropout is live); h part: this layer at tt - 1
            w.off_x = t.layer == 0 ? 0 : ((a.drop ? xd_base : 0) + ((t.layer - 1) * 2 + (tt & 1)) * slot_f * 4);
            w.off_h = (t.layer * 2 + ((tt - 1) & 1)) * slot_f * 4;
            if (w.n > 0) {
                if (ti == 0) ring_fill(ring0, rs, w);
                else ring_fill(ring1, rs, w);
            }
            // layer 0: the hoisted x*W_ih^T + biases of this step
            if (t.layer == 0 && cellw[ti] && row[ti] < B) {
                const float* g0 = a.gates + ((int64_t)tt * B + row[ti]) * 4 * H + 4 * t.unit + q;
#pragma unroll
                for (int r = 0; r < 4; ++r) pre_g[ti][r] = g0[r * H];
            }
        }
        // ---- products: branch-free rings (see WaveTask)
        if (on[0] && wt[0].n > 0) {
            acc[0] = ring_products(ring0, rs, wt[0], lds, lane);
            if (!cellw[0]) scratch[(0 * kWaves + wave) * 64 + lane] = acc[0];
        }
        if (on[1] && wt[1].n > 0) {
            acc[1] = ring_products(ring1, rs, wt[1], lds, lane);
            if (!cellw[1]) scratch[(1 * kWaves + wave) * 64 + lane] = acc[1];
        }
        MMQG_PSTAMP(1)
        __syncthreads();
        MMQG_PSTAMP(2)
        // ---- cell update by one wave per row block: task 0's by its first k-split wave, task 1's by its last one
        // (other waves, so the two cell updates of a workgroup run side by side)
        float st_gi[kMaxTasks], st_gf[kMaxTasks], st_gg[kMaxTasks], st_go[kMaxTasks], st_hd[kMaxTasks], st_y[kMaxTasks];
        bool st_on[kMaxTasks];
#pragma unroll
        for (int ti = 0; ti < kMaxTasks; ++ti) {
            const TaskInfo& t = tk[ti];
            const int tt = s - t.layer;
            st_on[ti] = false;
            if (!on[ti] || !cellw[ti]) continue;
            f32x4 sum = acc[ti];
            const int per = (t.nch + t.ksplit - 1) / t.ksplit;
            for (int k2 = 0; k2 < t.ksplit && k2 * per < t.nch; ++k2) {      // the other k-split waves that had chunks
                if (k2 == ks[ti]) continue;
                const f32x4 p = scratch[(ti * kWaves + mbi[ti] + k2 * t.msplit) * 64 + lane];
                sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
            }
            const int b = row[ti], u = 4 * t.unit + q, l = t.layer;
            const bool valid = b < B;
            const bool active = valid && tt < len[ti];
            const float gi = sigmoidf_(sum.x + pre_g[ti][0] + bias[ti][0]);
            const float gf = sigmoidf_(sum.y + pre_g[ti][1] + bias[ti][1]);
            const float gg = tanhf(sum.z + pre_g[ti][2] + bias[ti][2]);
            const float go = sigmoidf_(sum.w + pre_g[ti][3] + bias[ti][3]);
            if (active) {
                creg[ti] = gf * creg[ti] + gi * gg;
                hreg[ti] = go * tanhf(creg[ti]);
            }
            float hd = 0.f;
            const bool to_above = l < L - 1;
            if (a.drop && to_above && active)
                hd = hreg[ti] * dropout_scale(seed, a.stream_base + (uint64_t)l * T + tt, (uint64_t)((int64_t)b * H + u), a.drop_p);
            // exchange: h(t) for this layer's next step and for the layer above (the dropped copy when dropout is live).
            // The four hidden units of a row sit in lanes j, j+16, j+32, j+48: lane j collects them and stores 16 bytes
            // (a wave then writes one 256-byte run with 16 write-through requests instead of 64 four-byte ones).
            const int xoff = ((l * 2 + (tt & 1)) * slot_f) * 4 + ((t.unit * kRows + b) * 4) * 4;
            {
                const float hv = valid ? hreg[ti] : 0.f;
                u32x4 pk = {__float_as_uint(hv), __float_as_uint(__shfl(hv, j + 16, 64)), __float_as_uint(__shfl(hv, j + 32, 64)),
                            __float_as_uint(__shfl(hv, j + 48, 64))};
                if (q == 0) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, xoff, 0, 16);
            }
            if (a.drop && to_above) {
                u32x4 pk = {__float_as_uint(hd), __float_as_uint(__shfl(hd, j + 16, 64)), __float_as_uint(__shfl(hd, j + 32, 64)),
                            __float_as_uint(__shfl(hd, j + 48, 64))};
                if (q == 0) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, xd_base + xoff, 0, 16);
            }
            st_gi[ti] = active ? gi : 0.f; st_gf[ti] = active ? gf : 0.f; st_gg[ti] = active ? gg : 0.f; st_go[ti] = active ? go : 0.f;
            st_hd[ti] = hd; st_y[ti] = active ? hreg[ti] : 0.f;
            st_on[ti] = valid;
        }
        MMQG_PSTAMP(3)
        gb::arrive(bar);
        MMQG_PSTAMP(4)
        // ---- results that only later kernels read (saved activations for the backward pass, the value rows): after
        // the arrival, so their acknowledgement is not on the path between two diagonals
#pragma unroll
        for (int ti = 0; ti < kMaxTasks; ++ti) {
            if (!st_on[ti]) continue;
            const TaskInfo& t = tk[ti];
            const int tt = s - t.layer, b = row[ti], u = 4 * t.unit + q, l = t.layer;
            float* grow = a.gates + (((int64_t)l * T + tt) * B + b) * 4 * H + u;
            grow[0] = st_gi[ti]; grow[H] = st_gf[ti]; grow[2 * H] = st_gg[ti]; grow[3 * H] = st_go[ti];
            const int64_t e = (((int64_t)l * (T + 1) + tt + 1) * B + b) * H + u;
            a.hs[e] = hreg[ti]; a.cs[e] = creg[ti];
            if (a.drop && l < L - 1) a.hdrop[(((int64_t)l * T + tt) * B + b) * H + u] = st_hd[ti];
            if (l == L - 1 && a.y) a.y[(int64_t)tt * a.y_stride_t + (int64_t)b * a.y_stride_b + u] = st_y[ti];
        }
        ok = gb::wait(bar);
        MMQG_PSTAMP(5)
    }
    if (!ok) {            // a barrier timed out (not every workgroup resident, or the test hook): loud, not silent
        gb::report_failure(a.sticky_fail, a.host_fail);
        if (tid == 0) a.poison[0] = __builtin_nanf("");
    }
}
#undef MMQG_PSTAMP

// ------------------------------------------------------------------------------------ host
struct HostTask { int layer, unit, mb0, nmb; };

// Longest-processing-time style assignment of (unit, row blocks) to G workgroups: whole units of the
// costliest layers first, units that do not fit a workgroup's share are cut by row blocks.
bool build_plan(int L, int H, int B, int G, Plan& plan, int& max_lds_bytes) {
    const int nmb_all = ceil_div(B, 16), units = H / 4;
    struct Wg { int64_t load = 0; int ntask = 0; int lds = 0; HostTask t[kMaxTasks]; };
    std::vector<Wg> wg(G);
    int64_t total = 0;
    for (int l = 0; l < L; ++l) total += (int64_t)units * nmb_all * (l == 0 ? H : 2 * H);
    const int64_t target = (total + G - 1) / G;
    const int fixed_bytes = 64 * 16 + kMaxTasks * kWaves * 64 * 16;   // the zero chunk + the k-split waves' partial tiles
    std::vector<int> order;
    for (int l = 1; l < L; ++l) order.push_back(l);
    order.push_back(0);                                        // layer 0 has the shortest K: last
    for (int l : order) {
        const int K = l == 0 ? H : 2 * H, wbytes = 16 * K * 4;
        for (int u = 0; u < units; ++u) {
            int mb = 0;
            while (mb < nmb_all) {
                int best = -1;
                for (int g = 0; g < G; ++g) {
                    if (wg[g].ntask >= kMaxTasks || wg[g].lds + wbytes + fixed_bytes > kLdsBudget) continue;
                    if (best < 0 || wg[g].load < wg[best].load) best = g;
                }
                if (best < 0) return false;
                Wg& w = wg[best];
                // everything that is left of the unit if that overshoots the share by at most 12% (task slots are
                // scarce: two per workgroup), else as many row blocks as fit the share
                const int rest = nmb_all - mb;
                int take = rest;
                if (w.load + (int64_t)rest * K > target + target * 12 / 100) {
                    const int64_t room = target + target / 50 - w.load;
                    take = (int)std::min<int64_t>(rest, std::max<int64_t>(1, room / K));
                }
                w.t[w.ntask++] = HostTask{l, u, mb, take};
                w.load += (int64_t)take * K;
                w.lds += wbytes;
                mb += take;
            }
        }
    }
    max_lds_bytes = 0;
    for (int g = 0; g < kMaxWG; ++g)
        for (int i = 0; i < kMaxTasks; ++i) plan.task[g * kMaxTasks + i] = 0;
    for (int g = 0; g < G; ++g) {
        max_lds_bytes = std::max(max_lds_bytes, wg[g].lds + fixed_bytes);
        for (int i = 0; i < wg[g].ntask; ++i) {
            const HostTask& t = wg[g].t[i];
            plan.task[g * kMaxTasks + i] = (uint32_t)t.layer | ((uint32_t)t.unit << 2) | ((uint32_t)t.mb0 << 14) | ((uint32_t)t.nmb << 17);
        }
    }
    return true;
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct WsLayout { int64_t bar, hx, xd, sticky, total; };
WsLayout ws_layout(int L, int H) {
    WsLayout w;
    w.bar = 0;
    w.hx = align_up((int64_t)sizeof(gb::XBar), 256);
    w.xd = w.hx + (int64_t)L * 2 * kRows * H * 4;
    w.sticky = w.xd + (int64_t)std::max(L - 1, 1) * 2 * kRows * H * 4;   // failure counter: zeroed by the CALLER, once
    w.total = w.sticky + 256;
    return w;
}

}  // namespace

namespace mmqg {

static int g_persist_launches = 0;
int persist_launch_count() { return g_persist_launches; }
// diagnostics: per-(workgroup, diagonal) wall-clock stamps of the following persistent launches go to buf (null = off)
static unsigned long long* g_trace_buf = nullptr;
static int64_t g_trace_words = 0;
void persist_set_trace(unsigned long long* buf, int64_t words) { g_trace_buf = buf; g_trace_words = buf ? words : 0; }

bool lstm_persist_shape_ok(int T, int B, int L, int H) {
    static const bool off = [] { const char* e = getenv("MMQG_NO_PERSIST"); return e && atoi(e) != 0; }();
    if (off) return false;
    // worth it only for real time loops over wide layers: the weights must fit the LDS of the chip
    return T >= 2 && B >= 1 && B <= kRows && L >= 1 && L <= kMaxL && H >= 128 && H % 16 == 0 && H / 4 < 4096;
}

int64_t lstm_persist_ws_bytes(int T, int B, int L, int H) {
    if (!lstm_persist_shape_ok(T, B, L, H)) return 0;
    persist_runtime_prepare();           // pinned failure word + event: outside any stream capture
    return ws_layout(L, H).total;
}

// 0 = done, 1 = not eligible (the caller takes the launch-per-diagonal path), < 0 = error
int lstm_seq_fwd_persistent(const mmqg_lstm_seq& d, hipStream_t s) {
    if (!d.persist_ws || !lstm_persist_shape_ok(d.T, d.B, d.L, d.H)) return 1;
    const WsLayout wl = ws_layout(d.L, d.H);
    if (d.persist_ws_bytes < wl.total || !aligned16(d.persist_ws)) return 1;
    for (int l = 0; l < d.L; ++l)
        if (!aligned16(d.w_hh[l]) || (l > 0 && !aligned16(d.w_ih[l]))) return 1;
    const int cus = persist_usable_cus(s, false);      // (never beside a collective: the forward of a step follows its Adam)
    const int G = std::min(cus, kMaxWG);
    if (G < 64) return 1;
    Plan plan;
    int lds_bytes = 0;
    if (!build_plan(d.L, d.H, d.B, G, plan, lds_bytes)) return 1;
    static int attr_set = 0;      // 0 = not tried, 1 = ok, -1 = refused (fall back to launches for good)
    if (attr_set == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_fwd_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_fwd_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e != hipSuccess) (void)hipGetLastError();
        attr_set = e == hipSuccess ? 1 : -1;
    }
    if (attr_set < 0) return 1;
    {   // the grid is one workgroup per CU: the kernel must fit a CU with this much LDS (checked once per size)
        static int occ_lds = -1, occ_ok = 0;
        if (occ_lds != lds_bytes) {
            int nb = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &nb, reinterpret_cast<const void*>(lstm_persist_fwd_kernel<false>), kThreads, (size_t)lds_bytes);
            if (e != hipSuccess) (void)hipGetLastError();
            occ_lds = lds_bytes; occ_ok = (e == hipSuccess && nb >= 1) ? 1 : 0;
        }
        if (!occ_ok) return 1;
    }
    // no two persistent launches in flight on one device (persist_rt.hip): declined -> one launch per diagonal
    if (persist_begin(s) != 0) return 1;

    const int T = d.T, B = d.B, H = d.H, L = d.L;
    const int64_t BH = (int64_t)B * H;
    char* ws = reinterpret_cast<char*>(d.persist_ws);
    // initial states (slot 0 of hs / cs) and the zeroed barrier block: one launch
    CopySeg init[2 * MMQG_MAX_LAYERS + 1];
    for (int l = 0; l < L; ++l) {
        init[2 * l] = CopySeg{d.hs + (int64_t)l * (T + 1) * BH, d.h0 ? d.h0 + l * BH : nullptr, BH};
        init[2 * l + 1] = CopySeg{d.cs + (int64_t)l * (T + 1) * BH, d.c0 ? d.c0 + l * BH : nullptr, BH};
    }
    init[2 * L] = CopySeg{reinterpret_cast<float*>(ws + wl.bar), nullptr, (int64_t)(wl.hx / 4)};
    MMQG_TRY(copy_or_zero_multi(init, 2 * L + 1, s));
    // layer 0: every input product at once
    MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, T * B, 4 * H, d.In, d.x, d.ldx, d.w_ih[0], d.In, nullptr, 0, nullptr, 0, 0,
                      d.b_ih[0], d.b_hh[0], 0, d.gates, 4 * H, -1, s));
    PersistArgs a{};
    a.T = T; a.B = B; a.L = L; a.H = H; a.G = G;
    for (int l = 0; l < L; ++l) { a.w_ih[l] = d.w_ih[l]; a.w_hh[l] = d.w_hh[l]; a.b_ih[l] = d.b_ih[l]; a.b_hh[l] = d.b_hh[l]; }
    a.h0 = d.h0; a.c0 = d.c0; a.lens = d.lens;
    a.gates = d.gates; a.hs = d.hs; a.cs = d.cs; a.hdrop = d.hdrop; a.y = d.y; a.y_stride_t = d.y_stride_t; a.y_stride_b = d.y_stride_b;
    a.drop = (d.training && d.dropout_p > 0.f && L > 1) ? 1 : 0;
    a.drop_p = a.drop ? d.dropout_p : 0.f; a.seed = d.seed; a.stream_base = d.stream_base; a.seed_off = d.seed_offset;
    a.bar = reinterpret_cast<gb::XBar*>(ws + wl.bar);
    a.hx = reinterpret_cast<float*>(ws + wl.hx);
    a.xd = reinterpret_cast<float*>(ws + wl.xd);       // directly behind hx: one buffer descriptor covers both
    a.poison = d.hs + ((int64_t)(L - 1) * (T + 1) + T) * BH;        // final h of the top layer
    a.sticky_fail = reinterpret_cast<unsigned*>(ws + wl.sticky);
    a.host_fail = persist_host_fail_word();
    a.expect_wg = (unsigned)(G + persist_test_extra_wg());
    a.max_spins = persist_test_max_spins() ? persist_test_max_spins() : gb::kDefaultSpins;
    a.trace = nullptr;
    if (g_trace_buf && (int64_t)G * (T + L - 1) * 6 <= g_trace_words) a.trace = g_trace_buf;
    if (a.trace) hipLaunchKernelGGL(lstm_persist_fwd_kernel<true>, dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a, plan);
    else hipLaunchKernelGGL(lstm_persist_fwd_kernel<false>, dim3(G), dim3(kThreads), (size_t)lds_bytes, s, a, plan);
    g_persist_launches += 1;
    persist_end(s);
    return check_launch("lstm_persist_fwd");
}

}  // namespace mmqg
