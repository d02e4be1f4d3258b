// LSTM cell update and its gradient (gate order i,f,g,o as torch.nn.LSTM; reference call
// sites model/encoder.py:54,91 and model/decoder.py:69).  Pure element-wise work over
// [B][H]: HBM/L2-bound, one thread per hidden unit, consecutive lanes on consecutive units so
// every wave instruction moves one contiguous 256-byte segment.
#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

struct CellFwdK {
    int B, H;
    float* gates; int ld_g;
    const float* h_prev; const float* c_prev;
    float* h_out; float* c_out;
    float* h_drop;
    float* y_out; int64_t y_stride_b;
    const int32_t* lens; int t;
    float p; uint64_t seed; uint64_t stream_id; const int32_t* seed_off;
};

__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const int32_t* off) {
    return off ? seed + (uint64_t)(uint32_t)off[0] * 0x9E3779B97F4A7C15ull : seed;
}

__device__ __forceinline__ void cell_fwd_one(const CellFwdK& a, int b, int j, bool active) {
    float* g = a.gates + (int64_t)b * a.ld_g;
    const int64_t e = (int64_t)b * a.H + j;
    const float hp = a.h_prev[e], cp = a.c_prev[e];
    float h, c;
    if (active) {
        const float gi = sigmoidf_(g[j]);
        const float gf = sigmoidf_(g[a.H + j]);
        const float gg = tanhf(g[2 * a.H + j]);
        const float go = sigmoidf_(g[3 * a.H + j]);
        c = gf * cp + gi * gg;
        h = go * tanhf(c);
        g[j] = gi; g[a.H + j] = gf; g[2 * a.H + j] = gg; g[3 * a.H + j] = go;
    } else {
        c = cp; h = hp;
        g[j] = 0.f; g[a.H + j] = 0.f; g[2 * a.H + j] = 0.f; g[3 * a.H + j] = 0.f;
    }
    a.h_out[e] = h;
    a.c_out[e] = c;
    if (a.h_drop) a.h_drop[e] = active ? h * dropout_scale(eff_seed(a.seed, a.seed_off), a.stream_id, (uint64_t)e, a.p) : 0.f;
    if (a.y_out) a.y_out[(int64_t)b * a.y_stride_b + j] = active ? h : 0.f;
}

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(CellFwdK a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.B * a.H) return;
    const int b = (int)(idx / a.H), j = (int)(idx % a.H);
    const bool active = a.lens ? (a.t < a.lens[b]) : true;
    cell_fwd_one(a, b, j, active);
}

struct CellBwdK {
    int B, H;
    const float* gates_act;
    const float* c_prev; const float* c_new;
    float* dh_rec;
    const float* dh_above; int64_t above_stride_b;
    float p; uint64_t seed; uint64_t stream_id; const int32_t* seed_off;
    const float* dh_extra; int64_t extra_stride_b;
    float* dc;
    float* dgates; int ld_dg;
    const int32_t* lens; int t;
};

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(CellBwdK a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.B * a.H) return;
    const int b = (int)(idx / a.H), j = (int)(idx % a.H);
    const bool active = a.lens ? (a.t < a.lens[b]) : true;
    float* dg = a.dgates + (int64_t)b * a.ld_dg;
    const float dh_in = a.dh_rec[idx];
    if (!active) {
        // finished row: state was carried, so the gradient is carried too (dc stays as it is)
        dg[j] = 0.f; dg[a.H + j] = 0.f; dg[2 * a.H + j] = 0.f; dg[3 * a.H + j] = 0.f;
        return;   // dh_rec keeps dh_in: it is the pass-through part for step t-1
    }
    float dh = dh_in;
    if (a.dh_above) {
        float s = 1.f;
        if (a.p > 0.f) s = dropout_scale(eff_seed(a.seed, a.seed_off), a.stream_id, (uint64_t)idx, a.p);
        dh += a.dh_above[(int64_t)b * a.above_stride_b + j] * s;
    }
    if (a.dh_extra) dh += a.dh_extra[(int64_t)b * a.extra_stride_b + j];
    const float* g = a.gates_act + (int64_t)b * 4 * a.H;
    const float gi = g[j], gf = g[a.H + j], gg = g[2 * a.H + j], go = g[3 * a.H + j];
    const float tc = tanhf(a.c_new[idx]);
    const float dct = a.dc[idx] + dh * go * (1.f - tc * tc);
    dg[j] = dct * gg * gi * (1.f - gi);
    dg[a.H + j] = dct * a.c_prev[idx] * gf * (1.f - gf);
    dg[2 * a.H + j] = dct * gi * (1.f - gg * gg);
    dg[3 * a.H + j] = dh * tc * go * (1.f - go);
    a.dc[idx] = dct * gf;
    a.dh_rec[idx] = 0.f;   // the recurrent part dgates*W_hh is added by the following GEMM
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float* out, int64_t n, float p, uint64_t seed,
                                                           uint64_t stream_id, const int32_t* seed_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = p > 0.f ? dropout_scale(eff_seed(seed, seed_off), stream_id, (uint64_t)i, p) : 1.f;
}

}  // namespace

namespace mmqg {

int lstm_cell_fwd(const CellFwd& f, hipStream_t s) {
    MMQG_REQUIRE(f.B >= 0 && f.H > 0, "lstm_cell_fwd: bad shape");
    if (f.B == 0) return 0;
    MMQG_REQUIRE(f.gates && f.h_prev && f.c_prev && f.h_out && f.c_out, "lstm_cell_fwd: null pointer");
    MMQG_REQUIRE(f.ld_g >= 4 * f.H, "lstm_cell_fwd: ld_g < 4H");
    MMQG_REQUIRE(f.p >= 0.f && f.p < 1.f, "lstm_cell_fwd: dropout p must be in [0,1)");
    CellFwdK k{f.B, f.H, f.gates, f.ld_g, f.h_prev, f.c_prev, f.h_out, f.c_out, f.h_drop,
               f.y_out, f.y_stride_b, f.lens, f.t, f.p, f.seed, f.stream_id, f.seed_off};
    const int64_t n = (int64_t)f.B * f.H;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, k);
    return check_launch("lstm_cell_fwd");
}

int lstm_cell_bwd(const CellBwd& f, hipStream_t s) {
    MMQG_REQUIRE(f.B >= 0 && f.H > 0, "lstm_cell_bwd: bad shape");
    if (f.B == 0) return 0;
    MMQG_REQUIRE(f.gates_act && f.c_prev && f.c_new && f.dh_rec && f.dc && f.dgates, "lstm_cell_bwd: null pointer");
    MMQG_REQUIRE(f.ld_dg >= 4 * f.H, "lstm_cell_bwd: ld_dg < 4H");
    MMQG_REQUIRE(!f.dh_pre, "lstm_cell_bwd: dh_pre is a fused-kernel input");
    CellBwdK k{f.B, f.H, f.gates_act, f.c_prev, f.c_new, f.dh_rec, f.dh_above, f.above_stride_b,
               f.p, f.seed, f.stream_id, f.seed_off, f.dh_extra, f.extra_stride_b, f.dc, f.dgates, f.ld_dg, f.lens, f.t};
    const int64_t n = (int64_t)f.B * f.H;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, k);
    return check_launch("lstm_cell_bwd");
}

int dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t stream_id, const int32_t* seed_off,
                 hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && out, "dropout_mask: bad arguments");
    MMQG_REQUIRE(p >= 0.f && p < 1.f, "dropout_mask: p must be in [0,1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, out, n, p, seed,
                       stream_id, seed_off);
    return check_launch("dropout_mask");
}

}  // namespace mmqg
