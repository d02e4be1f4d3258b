// Whole-sequence executors: enqueue every kernel of a time loop from C++ so the host pays one
// FFI call (or one hipGraph replay) per sequence instead of one Python round trip per token,
// which is how the reference drives its modules (train.py:164-175).
//
// Scheduling idea (both executors): anything that does not depend on the recurrence is
// hoisted out of the time loop into one large GEMM over all T*B rows — the input products
// X*W_ih^T of a layer, the embedded-word part of the attention scores and of the decoder's
// layer-0 gates, every weight gradient (dW = dGates^T * X over all steps at once) and the
// gradient of the value tensors.  The loop itself keeps only h*W_hh^T (+ the attention
// context product in the decoder), the cell update and the attention kernels.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

using namespace mmqg;

int copy_or_zero(float* dst, const float* src, size_t n, hipStream_t s) {
    return copy_or_zero_f32(dst, src, (int64_t)n, s);
}

int check_lstm(const mmqg_lstm_seq& d, const char* who) {
    MMQG_REQUIRE(d.T >= 0 && d.B >= 0 && d.H > 0 && d.In > 0, "%s: bad shape", who);
    MMQG_REQUIRE(d.L >= 1 && d.L <= MMQG_MAX_LAYERS, "%s: L must be in [1,%d]", who, MMQG_MAX_LAYERS);
    MMQG_REQUIRE(d.x && d.ldx >= d.In, "%s: bad input", who);
    MMQG_REQUIRE(d.gates && d.hs && d.cs, "%s: null state buffer", who);
    for (int l = 0; l < d.L; ++l)
        MMQG_REQUIRE(d.w_ih[l] && d.w_hh[l] && d.b_ih[l] && d.b_hh[l], "%s: null weight (layer %d)", who, l);
    MMQG_REQUIRE(d.dropout_p >= 0.f && d.dropout_p < 1.f, "%s: dropout_p must be in [0,1)", who);
    const bool drop = d.training && d.dropout_p > 0.f && d.L > 1;
    MMQG_REQUIRE(!drop || d.hdrop, "%s: training with dropout needs the hdrop buffer", who);
    return 0;
}

inline bool lstm_drop(const mmqg_lstm_seq& d) { return d.training && d.dropout_p > 0.f && d.L > 1; }

// the wide backward / plain kernel's workspace is the caller's: handed to skinny.hip for the duration of one executor call
struct WideWsScope {
    WideWsScope(float* ws, int64_t bytes) { mmqg::skinny_set_wide_ws(ws, bytes); }
    ~WideWsScope() { mmqg::skinny_set_wide_ws(nullptr, 0); }
};

// MMQG_NO_WAVEFRONT=1: layer-by-layer time loops (one fused launch per layer-step) for A/B comparisons
inline bool g_no_wavefront() {
    static const bool v = [] { const char* e = getenv("MMQG_NO_WAVEFRONT"); return e && atoi(e) != 0; }();
    return v;
}

// MMQG_NO_FUSE=1 keeps the round-1 path (tiled GEMM + separate cell kernels) for A/B comparisons
inline bool g_no_fuse() {
    static const bool v = [] { const char* e = getenv("MMQG_NO_FUSE"); return e && atoi(e) != 0; }();
    return v;
}

}  // namespace

namespace mmqg {

// Forward of a layer stack as a wavefront: diagonal s runs layer l at time t = s - l for every l,
// up to three independent layer-steps in ONE launch (T + L - 1 dependent launches instead of L*T).
// Layers >= 1 take their input product in the same launch (x*W_ih^T + h*W_hh^T + biases).
static int lstm_seq_fwd_wavefront(const mmqg_lstm_seq& d, hipStream_t s) {
    const int T = d.T, B = d.B, H = d.H, L = d.L;
    const int64_t BH = (int64_t)B * H, G = (int64_t)B * 4 * H;
    const bool drop = lstm_drop(d);
    CopySeg init[2 * MMQG_MAX_LAYERS];      // initial states of the whole stack: one launch
    for (int l = 0; l < L; ++l) {
        init[2 * l] = CopySeg{d.hs + (int64_t)l * (T + 1) * BH, d.h0 ? d.h0 + l * BH : nullptr, BH};
        init[2 * l + 1] = CopySeg{d.cs + (int64_t)l * (T + 1) * BH, d.c0 ? d.c0 + l * BH : nullptr, BH};
    }
    MMQG_TRY(copy_or_zero_multi(init, 2 * L, s));
    // layer 0: every input product at once
    MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, T * B, 4 * H, d.In, d.x, d.ldx, d.w_ih[0], d.In, nullptr, 0, nullptr, 0, 0,
                      d.b_ih[0], d.b_hh[0], 0, d.gates, 4 * H, -1, s));
    for (int diag = 0; diag < T + L - 1; ++diag) {
        SkinnyFwdJob jobs[3];
        int nj = 0;
        for (int l = 0; l < L; ++l) {
            const int t = diag - l;
            if (t < 0 || t >= T) continue;
            float* hs_l = d.hs + (int64_t)l * (T + 1) * BH;
            float* cs_l = d.cs + (int64_t)l * (T + 1) * BH;
            SkinnyFwdJob& j = jobs[nj++];
            j = SkinnyFwdJob{};
            if (l == 0) {
                j.pairs[0] = SkinnyPair{hs_l + t * BH, H, d.w_hh[0], H, H, 0};
                j.npairs = 1; j.gates_has_pre = 1;
            } else {
                const float* xin = drop ? d.hdrop + (int64_t)(l - 1) * T * BH + t * BH
                                        : d.hs + (int64_t)(l - 1) * (T + 1) * BH + (t + 1) * BH;
                j.pairs[0] = SkinnyPair{xin, H, d.w_ih[l], H, H, 0};
                j.pairs[1] = SkinnyPair{hs_l + t * BH, H, d.w_hh[l], H, H, 0};
                j.npairs = 2; j.gates_has_pre = 0; j.bias1 = d.b_ih[l]; j.bias2 = d.b_hh[l];
            }
            CellFwd& c = j.cell;
            c.B = B; c.H = H; c.gates = d.gates + (int64_t)l * T * G + t * G; c.ld_g = 4 * H;
            c.h_prev = hs_l + t * BH; c.c_prev = cs_l + t * BH;
            c.h_out = hs_l + (t + 1) * BH; c.c_out = cs_l + (t + 1) * BH;
            c.h_drop = (drop && l < L - 1) ? d.hdrop + (int64_t)l * T * BH + t * BH : nullptr;
            c.y_out = (l == L - 1 && d.y) ? d.y + t * d.y_stride_t : nullptr;
            c.y_stride_b = d.y_stride_b;
            c.lens = d.lens; c.t = t;
            c.p = drop ? d.dropout_p : 0.f; c.seed = d.seed; c.seed_off = d.seed_offset; c.stream_id = d.stream_base + (uint64_t)l * T + t;
            if (nj == 3 || l == L - 1) { MMQG_TRY(skinny_cell_fwd_multi(jobs, nj, s)); nj = 0; }
        }
        if (nj > 0) MMQG_TRY(skinny_cell_fwd_multi(jobs, nj, s));
    }
    return 0;
}

static bool lstm_wavefront_fwd_ok(const mmqg_lstm_seq& d) {
    if (d.L < 2 || d.T < 1 || d.H % 4 != 0 || g_no_fuse() || g_no_wavefront()) return false;
    const int H = d.H;
    for (int l = 0; l < d.L; ++l) {
        const SkinnyPair a{d.hs, H, d.w_hh[l], H, H, 0};
        if (!skinny_usable(&a, 1)) return false;
        if (l > 0) {
            const SkinnyPair b{d.hs, H, d.w_ih[l], H, H, 0};
            if (!skinny_usable(&b, 1)) return false;
            if (lstm_drop(d) && !mmqg::aligned16(d.hdrop)) return false;
        }
    }
    return true;
}

int lstm_seq_fwd(const mmqg_lstm_seq& d, hipStream_t s) {
    MMQG_TRY(check_lstm(d, "lstm_seq_fwd"));
    if (d.B == 0) return 0;
    if (d.persist_ws && !g_no_fuse()) {      // the whole time loop as one persistent launch (persist.hip)
        const int rc = lstm_seq_fwd_persistent(d, s);
        if (rc <= 0) return rc;
    }
    if (lstm_wavefront_fwd_ok(d)) return lstm_seq_fwd_wavefront(d, s);
    const int T = d.T, B = d.B, H = d.H, L = d.L;
    const int64_t BH = (int64_t)B * H, G = (int64_t)B * 4 * H;
    const bool drop = lstm_drop(d);
    for (int l = 0; l < L; ++l) {
        float* hs_l = d.hs + (int64_t)l * (T + 1) * BH;
        float* cs_l = d.cs + (int64_t)l * (T + 1) * BH;
        MMQG_TRY(copy_or_zero(hs_l, d.h0 ? d.h0 + l * BH : nullptr, (size_t)BH, s));
        MMQG_TRY(copy_or_zero(cs_l, d.c0 ? d.c0 + l * BH : nullptr, (size_t)BH, s));
        if (T == 0) continue;
        const float* X; int ldx, in;
        if (l == 0) { X = d.x; ldx = d.ldx; in = d.In; }
        else { X = drop ? d.hdrop + (int64_t)(l - 1) * T * BH : d.hs + (int64_t)(l - 1) * (T + 1) * BH + BH; ldx = H; in = H; }
        float* gates_l = d.gates + (int64_t)l * T * G;
        // all input products of the layer at once: gates[t] = X[t]*W_ih^T + b_ih + b_hh
        MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, T * B, 4 * H, in, X, ldx, d.w_ih[l], in, nullptr, 0, nullptr, 0, 0,
                          d.b_ih[l], d.b_hh[l], 0, gates_l, 4 * H, -1, s));
        const SkinnyPair probe{hs_l, H, d.w_hh[l], H, H, 0};
        const bool fused = (H % 4 == 0) && skinny_usable(&probe, 1) && !g_no_fuse();
        for (int t = 0; t < T; ++t) {
            if (!fused)
                MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, B, 4 * H, H, hs_l + t * BH, H, d.w_hh[l], H, nullptr, 0,
                                  nullptr, 0, 0, nullptr, nullptr, 1, gates_l + t * G, 4 * H, -1, s));
            CellFwd c{};
            c.B = B; c.H = H; c.gates = gates_l + t * G; c.ld_g = 4 * H;
            c.h_prev = hs_l + t * BH; c.c_prev = cs_l + t * BH;
            c.h_out = hs_l + (t + 1) * BH; c.c_out = cs_l + (t + 1) * BH;
            c.h_drop = (drop && l < L - 1) ? d.hdrop + (int64_t)l * T * BH + t * BH : nullptr;
            c.y_out = (l == L - 1 && d.y) ? d.y + t * d.y_stride_t : nullptr;
            c.y_stride_b = d.y_stride_b;
            c.lens = d.lens; c.t = t;
            c.p = drop ? d.dropout_p : 0.f; c.seed = d.seed; c.seed_off = d.seed_offset; c.stream_id = d.stream_base + (uint64_t)l * T + t;
            if (fused) {
                // one launch: gates[t] (hoisted X*W_ih^T + biases) += h(t-1)*W_hh^T, then the cell update
                const SkinnyPair pr{hs_l + t * BH, H, d.w_hh[l], H, H, 0};
                MMQG_TRY(skinny_cell_fwd(&pr, 1, 1, nullptr, nullptr, c, s));
            } else {
                MMQG_TRY(lstm_cell_fwd(c, s));
            }
        }
    }
    return 0;
}

// Backward time loops of a layer stack as a wavefront over (layer, time) anti-diagonals: layer l at
// time t needs dgates_l(t+1) (recurrent) and dgates_{l+1}(t) (gradient of its output as the input of
// the layer above, through this layer's dropout mask) — both from the previous diagonal.
static int lstm_seq_bwd_wavefront(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, hipStream_t s) {
    const int T = d.T, B = d.B, H = d.H, L = d.L;
    const int64_t BH = (int64_t)B * H, G = (int64_t)B * 4 * H;
    const bool drop = lstm_drop(d);
    CopySeg init[2 * MMQG_MAX_LAYERS];
    for (int l = 0; l < L; ++l) {
        init[2 * l] = CopySeg{g.dh + l * BH, g.dhT ? g.dhT + l * BH : nullptr, BH};
        init[2 * l + 1] = CopySeg{g.dc + l * BH, g.dcT ? g.dcT + l * BH : nullptr, BH};
    }
    MMQG_TRY(copy_or_zero_multi(init, 2 * L, s));
    for (int diag = 0; diag < T + L - 1; ++diag) {
        SkinnyBwdJob jobs[3];
        int nj = 0;
        for (int l = L - 1; l >= 0; --l) {
            const int t = (T - 1) - (diag - (L - 1 - l));
            if (t < 0 || t >= T) continue;
            const float* cs_l = d.cs + (int64_t)l * (T + 1) * BH;
            SkinnyBwdJob j{};
            if (t < T - 1)
                j.pairs[j.npairs++] = SkinnyPair{g.dgates + (int64_t)l * T * G + (t + 1) * G, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
            if (l < L - 1)
                j.pairs[j.npairs++] = SkinnyPair{g.dgates + (int64_t)(l + 1) * T * G + t * G, 4 * H, d.w_ihT[l + 1], 4 * H, 4 * H, 1};
            CellBwd& c = j.cell;
            c.B = B; c.H = H; c.gates_act = d.gates + (int64_t)l * T * G + t * G;
            c.c_prev = cs_l + t * BH; c.c_new = cs_l + (t + 1) * BH;
            c.dh_rec = g.dh + l * BH;
            if (l < L - 1) {
                c.p = drop ? d.dropout_p : 0.f; c.seed = d.seed; c.seed_off = d.seed_offset; c.stream_id = d.stream_base + (uint64_t)l * T + t;
            } else if (g.dy) {
                c.dh_extra = g.dy + t * g.dy_stride_t; c.extra_stride_b = g.dy_stride_b;
            }
            c.dc = g.dc + l * BH; c.dgates = g.dgates + (int64_t)l * T * G + t * G; c.ld_dg = 4 * H;
            c.lens = d.lens; c.t = t;
            if (j.npairs == 0) { MMQG_TRY(lstm_cell_bwd(c, s)); continue; }     // top layer, last step
            jobs[nj++] = j;
            if (nj == 3) { MMQG_TRY(skinny_cell_bwd_multi(jobs, nj, s)); nj = 0; }
        }
        if (nj > 0) MMQG_TRY(skinny_cell_bwd_multi(jobs, nj, s));
    }
    if (g.dh0 || g.dc0) {
        for (int l = 0; l < L; ++l) {
            if (g.dh0) {   // gradient of the initial state: carry + dgates_l(0) * W_hh_l
                const SkinnyPair pr{g.dgates + (int64_t)l * T * G, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
                MMQG_TRY(skinny_plain(B, H, &pr, 1, nullptr, 1, g.dh + l * BH, H, s));
                MMQG_TRY(copy_or_zero(g.dh0 + l * BH, g.dh + l * BH, (size_t)BH, s));
            }
            if (g.dc0) MMQG_TRY(copy_or_zero(g.dc0 + l * BH, g.dc + l * BH, (size_t)BH, s));
        }
    }
    return 0;
}

static bool lstm_wavefront_bwd_ok(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g) {
    if (d.L < 2 || d.L > 3 || g_no_fuse() || g_no_wavefront()) return false;
    const int H = d.H;
    for (int l = 0; l < d.L; ++l) {
        const SkinnyPair a{g.dgates, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
        if (!d.w_hhT[l] || !skinny_usable(&a, 1)) return false;
        if (l > 0) {
            const SkinnyPair b{g.dgates, 4 * H, d.w_ihT[l], 4 * H, 4 * H, 1};
            if (!d.w_ihT[l] || !skinny_usable(&b, 1)) return false;
        }
    }
    return true;
}

// gradient of the initial state after the persistent backward: what the kernel left in dh (the carry) + dgates_l(0) W_hh_l
static int initial_state_grads_after_persistent(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, hipStream_t s) {
    const int T = d.T, B = d.B, H = d.H, L = d.L;
    const int64_t BH = (int64_t)B * H, G = (int64_t)B * 4 * H;
    for (int l = 0; l < L && (g.dh0 || g.dc0); ++l) {
        if (g.dh0) {
            MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, B, H, 4 * H, g.dgates + (int64_t)l * T * G, 4 * H, d.w_hh[l], H,
                              nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 1, g.dh + l * BH, H, -1, s));
            MMQG_TRY(copy_or_zero(g.dh0 + l * BH, g.dh + l * BH, (size_t)BH, s));
        }
        if (g.dc0) MMQG_TRY(copy_or_zero(g.dc0 + l * BH, g.dc + l * BH, (size_t)BH, s));
    }
    return 0;
}

int lstm_seq_bwd(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, hipStream_t s);

// The backward TIME LOOPS (phase 1) of two independent stacks: the second one a single layer of the same batch and
// width (the frame LSTM beside the text encoder).  ONE persistent launch when persist_bwd.hip takes the pair, otherwise
// each stack's own loop (same results).  The callers run phase 2 (weight gradients, dx) of each afterwards.
int lstm_seq_bwd_pair(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, const mmqg_lstm_seq& d2, const mmqg_lstm_seq_grad& g2,
                      hipStream_t s) {
    MMQG_TRY(check_lstm(d, "lstm_seq_bwd_pair"));
    MMQG_TRY(check_lstm(d2, "lstm_seq_bwd_pair (second stack)"));
    if (d.B > 0 && d.T > 0 && d2.B > 0 && d2.T > 0 && g.persist_ws && !g_no_fuse() && g.dgates && g.dh && g.dc && g2.dgates &&
        g2.dh && g2.dc) {
        const int rc = lstm_seq_bwd_persistent(d, g, &d2, &g2, s);
        if (rc < 0) return rc;
        if (rc == 0) {
            MMQG_TRY(initial_state_grads_after_persistent(d, g, s));
            return initial_state_grads_after_persistent(d2, g2, s);
        }
    }
    mmqg_lstm_seq_grad a = g, b = g2;
    a.phase = 1; b.phase = 1;
    MMQG_TRY(lstm_seq_bwd(d, a, s));
    return lstm_seq_bwd(d2, b, s);
}

int lstm_seq_bwd(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, hipStream_t s) {
    MMQG_TRY(check_lstm(d, "lstm_seq_bwd"));
    if (d.B == 0 || d.T == 0) return 0;
    MMQG_REQUIRE(g.dgates && g.dh && g.dc, "lstm_seq_bwd: null scratch buffer");
    MMQG_REQUIRE(d.L == 1 || g.dxl, "lstm_seq_bwd: multi-layer backward needs dxl");
    MMQG_REQUIRE(!g.dx || g.lddx >= d.In, "lstm_seq_bwd: lddx < In");
    const int T = d.T, B = d.B, H = d.H, L = d.L;
    const int64_t BH = (int64_t)B * H, G = (int64_t)B * 4 * H;
    const bool drop = lstm_drop(d);
    MMQG_REQUIRE(g.phase >= 0 && g.phase <= 2, "lstm_seq_bwd: phase must be 0, 1 or 2");
    const WideWsScope wide_scope(g.wide_ws, g.wide_ws_bytes);
    const bool do_wgrad = g.phase != 1;
    bool do_loop = g.phase != 2;
    if (do_loop && g.persist_ws && !g_no_fuse()) {      // all anti-diagonals as one persistent launch (persist_bwd.hip)
        const int rc = lstm_seq_bwd_persistent(d, g, nullptr, nullptr, s);
        if (rc < 0) return rc;
        if (rc == 0) {
            do_loop = false;
            MMQG_TRY(initial_state_grads_after_persistent(d, g, s));
        }
    }
    if (do_loop && lstm_wavefront_bwd_ok(d, g)) {
        MMQG_TRY(lstm_seq_bwd_wavefront(d, g, s));
        do_loop = false;
    }
    for (int l = L - 1; l >= 0 && do_loop; --l) {
        const float* cs_l = d.cs + (int64_t)l * (T + 1) * BH;
        const float* gates_l = d.gates + (int64_t)l * T * G;
        float* dg_l = g.dgates + (int64_t)l * T * G;
        MMQG_TRY(copy_or_zero(g.dh, g.dhT ? g.dhT + l * BH : nullptr, (size_t)BH, s));
        MMQG_TRY(copy_or_zero(g.dc, g.dcT ? g.dcT + l * BH : nullptr, (size_t)BH, s));
        const SkinnyPair probe{dg_l, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
        const bool fused = d.w_hhT[l] && skinny_usable(&probe, 1) && !g_no_fuse();
        for (int t = T - 1; t >= 0; --t) {
            CellBwd c{};
            c.B = B; c.H = H; c.gates_act = gates_l + t * G;
            c.c_prev = cs_l + t * BH; c.c_new = cs_l + (t + 1) * BH;
            c.dh_rec = g.dh;
            if (l < L - 1) {
                c.dh_above = g.dxl + t * BH; c.above_stride_b = H;
                c.p = drop ? d.dropout_p : 0.f; c.seed = d.seed; c.seed_off = d.seed_offset; c.stream_id = d.stream_base + (uint64_t)l * T + t;
            } else if (g.dy) {
                c.dh_extra = g.dy + t * g.dy_stride_t; c.extra_stride_b = g.dy_stride_b;
            }
            c.dc = g.dc; c.dgates = dg_l + t * G; c.ld_dg = 4 * H;
            c.lens = d.lens; c.t = t;
            if (fused && t < T - 1) {
                // one launch: dh(t) = dgates(t+1) * W_hh + carried/extra gradients, then the cell backward
                const SkinnyPair pr{dg_l + (t + 1) * G, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
                MMQG_TRY(skinny_cell_bwd(&pr, 1, c, s));
            } else {
                MMQG_TRY(lstm_cell_bwd(c, s));
            }
            if (!fused)   // dh(t-1) += dgates(t) * W_hh
                MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, B, H, 4 * H, dg_l + t * G, 4 * H, d.w_hh[l], H, nullptr, 0,
                                  nullptr, 0, 0, nullptr, nullptr, 1, g.dh, H, -1, s));
        }
        if (fused && (g.dh0 != nullptr)) {   // gradient of the initial state: carry + dgates(0) * W_hh
            const SkinnyPair pr{dg_l, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
            MMQG_TRY(skinny_plain(B, H, &pr, 1, nullptr, 1, g.dh, H, s));
        }
        if (g.dh0) MMQG_TRY(copy_or_zero(g.dh0 + l * BH, g.dh, (size_t)BH, s));
        if (g.dc0) MMQG_TRY(copy_or_zero(g.dc0 + l * BH, g.dc, (size_t)BH, s));
        // gradient wrt the layer input, all steps at once: the next (lower) layer's loop consumes it
        if (l > 0)
            MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, T * B, H, 4 * H, dg_l, 4 * H, d.w_ih[l], H, nullptr, 0, nullptr,
                              0, 0, nullptr, nullptr, 0, g.dxl, H, -1, s));
    }
    if (do_wgrad) {
        // recurrence-free products over all T*B rows: layer-0 input gradient, then every weight gradient of the
        // stack as ONE grouped launch (dW += dGates^T X), bias gradients (b_ih and b_hh get the same sums)
        GemmProblem wg[2 * MMQG_MAX_LAYERS];
        float* cs1[2 * MMQG_MAX_LAYERS] = {};
        float* cs2[2 * MMQG_MAX_LAYERS] = {};
        int nw = 0;
        for (int l = L - 1; l >= 0; --l) {
            const float* hs_l = d.hs + (int64_t)l * (T + 1) * BH;
            const float* dg_l = g.dgates + (int64_t)l * T * G;
            const float* X; int ldx, in;
            if (l == 0) { X = d.x; ldx = d.ldx; in = d.In; }
            else { X = drop ? d.hdrop + (int64_t)(l - 1) * T * BH : d.hs + (int64_t)(l - 1) * (T + 1) * BH + BH; ldx = H; in = H; }
            if (l == 0 && g.dx)
                MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, T * B, in, 4 * H, dg_l, 4 * H, d.w_ih[0], in, nullptr, 0,
                                  nullptr, 0, 0, nullptr, nullptr, 0, g.dx, g.lddx, -1, s));
            if (g.dw_ih[l]) wg[nw++] = GemmProblem{4 * H, in, T * B, dg_l, 4 * H, X, ldx, g.dw_ih[l], in, 1};
            const int before = nw;
            if (g.dw_hh[l]) wg[nw++] = GemmProblem{4 * H, H, T * B, dg_l, 4 * H, hs_l, H, g.dw_hh[l], H, 1};
            // bias gradients = column sums of the gate gradients: they ride in one of the layer's weight-gradient
            // products (its staging pass reads every dG element anyway), or get a sweep of their own
            float* b1 = g.db_ih[l] ? g.db_ih[l] : g.db_hh[l];
            float* b2 = g.db_ih[l] ? g.db_hh[l] : nullptr;
            if (b1 && nw > before) { cs1[nw - 1] = b1; cs2[nw - 1] = b2; }
            else if (b1) MMQG_TRY(colsum_add2(dg_l, 4 * H, T * B, 4 * H, b1, b2, s));
        }
        MMQG_TRY(gemm_f32_wgrad_group(wg, cs1, cs2, nw, s));
    }
    return 0;
}

// ------------------------------------------------------------------------------- decoder
static int check_decoder(const mmqg_decoder_seq& d, const char* who) {
    MMQG_REQUIRE(d.T >= 0 && d.B >= 0 && d.H > 0 && d.E > 0, "%s: bad shape", who);
    MMQG_REQUIRE(d.L >= 1 && d.L <= MMQG_MAX_LAYERS, "%s: L must be in [1,%d]", who, MMQG_MAX_LAYERS);
    MMQG_REQUIRE(d.values.B == d.B && d.values.H > 0, "%s: values.B must equal B", who);
    MMQG_REQUIRE(d.xemb && d.w_attn && d.b_attn && d.h0 && d.c0, "%s: null input", who);
    MMQG_REQUIRE(d.scores && d.attn && d.ctx && d.gates && d.hs && d.cs, "%s: null buffer", who);
    MMQG_REQUIRE(d.ld_attn >= d.values.Lt + 2 * d.values.Lav, "%s: ld_attn too small", who);
    for (int l = 0; l < d.L; ++l)
        MMQG_REQUIRE(d.w_ih[l] && d.w_hh[l] && d.b_ih[l] && d.b_hh[l], "%s: null weight (layer %d)", who, l);
    MMQG_REQUIRE(d.dropout_p >= 0.f && d.dropout_p < 1.f, "%s: dropout_p must be in [0,1)", who);
    const bool drop = d.training && d.dropout_p > 0.f && d.L > 1;
    MMQG_REQUIRE(!drop || d.hdrop, "%s: training with dropout needs the hdrop buffer", who);
    return 0;
}

int decoder_seq_fwd(const mmqg_decoder_seq& d, hipStream_t s) {
    MMQG_TRY(check_decoder(d, "decoder_seq_fwd"));
    if (d.B == 0) return 0;
    const int T = d.T, B = d.B, H = d.H, L = d.L, E = d.E;
    const mmqg_attn_values& v = d.values;
    const int S = v.Lt + 2 * v.Lav, C = v.H + v.Da + v.Dv, Q = E + H, In0 = E + C;
    const int ldS = d.ld_attn;
    const int64_t BH = (int64_t)B * H, G = (int64_t)B * 4 * H;
    const bool drop = d.training && d.dropout_p > 0.f && L > 1;
    MMQG_REQUIRE(d.phase >= 0 && d.phase <= 2, "decoder_seq_fwd: phase must be 0, 1 or 2");
    if (d.phase != 1) {
        CopySeg init[2 * MMQG_MAX_LAYERS];
        for (int l = 0; l < L; ++l) {
            const int64_t hstr = d.h0_stride_l ? d.h0_stride_l : BH;
            init[2 * l] = CopySeg{d.hs + (int64_t)l * (T + 1) * BH, d.h0 + l * hstr, BH};
            init[2 * l + 1] = CopySeg{d.cs + (int64_t)l * (T + 1) * BH, d.c0 + l * hstr, BH};
        }
        MMQG_TRY(copy_or_zero_multi(init, 2 * L, s));
    }
    if (T == 0) return 0;
    if (d.phase != 2) {
        // hoisted: embedded-word part of the scores (+ bias) and of the layer-0 gates (+ b_ih + b_hh)
        MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, T * B, S, E, d.xemb, E, d.w_attn, Q, nullptr, 0, nullptr, 0, 0,
                          d.b_attn, nullptr, 0, d.scores, ldS, -1, s));
        MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, T * B, 4 * H, E, d.xemb, E, d.w_ih[0], In0, nullptr, 0, nullptr, 0, 0,
                          d.b_ih[0], d.b_hh[0], 0, d.gates, 4 * H, -1, s));
    }
    if (d.phase == 1) return 0;
    if (d.persist_ws && !g_no_fuse()) {      // the whole time loop as one persistent launch (persist_dec.hip)
        const int rc = decoder_seq_fwd_persistent(d, s);
        if (rc <= 0) return rc;
    }
    const float* htop_base = d.hs + (int64_t)(L - 1) * (T + 1) * BH;
    // Look-ahead (as in the backward loop): the recurrent half of a layer-step, h_l(t) W_hh_l^T, only needs h_l(t), which
    // exists one launch after cell (l, t) — a whole token before cell (l, t+1) uses it.  It is formed as an extra plain job
    // of the NEXT launch on the dependent chain (cell (l+1, t); for the top layer the score product of token t+1) straight
    // into the pre-activation slot gates_l[t+1], so every cell launch on the chain carries only the operand pair that
    // really is late (layer 0: the contexts, K 1664 -> 1152; layers >= 1: the layer below, K 1024 -> 512).
    // MEASURED (round 3, config 2): decoder forward 1.23 ms with it against 1.18 ms without — unlike the backward loop (K
    // 4096 -> 2048 per cell launch, 14.9 -> 9.3 us) the forward launches are not shortened by carrying fewer k-chunks
    // (their time is arguments + cold operands + epilogue + drain), and the extra jobs cost more than they save.  Kept
    // as an opt-in A/B switch (MMQG_AHEAD_FWD=1), parity-tested.
    static const bool ahead_fwd = [] { const char* e = getenv("MMQG_AHEAD_FWD"); return e && atoi(e) != 0; }();
    bool ahead = ahead_fwd && !g_no_fuse() && (H % 4 == 0) && B <= 64;          // (B > 64: the wide cell kernel, no extra jobs)
    for (int l = 0; l < L && ahead; ++l) {
        const SkinnyPair a{d.hs, H, d.w_hh[l], H, H, 0};
        const SkinnyPair b = l == 0 ? SkinnyPair{d.ctx, C, d.w_ih[0] + E, In0, C, 0} : SkinnyPair{d.hs, H, d.w_ih[l], H, H, 0};
        ahead = skinny_usable(&a, 1) && skinny_usable(&b, 1) && (!drop || l == 0 || aligned16(d.hdrop));
    }
    auto rec_job = [&](int l, int t_next) {       // gates_l[t_next] (+)= h_l(t_next - 1) W_hh_l^T
        SkinnyPlainJob j{};
        j.M = B; j.N = 4 * H; j.npairs = 1; j.beta = l == 0 ? 1 : 0;      // layer 0: onto the hoisted embedded-word part + biases
        j.C = d.gates + (int64_t)l * T * G + (int64_t)t_next * G; j.ldc = 4 * H;
        j.pairs[0] = SkinnyPair{d.hs + (int64_t)l * (T + 1) * BH + (int64_t)t_next * BH, H, d.w_hh[l], H, H, 0};
        return j;
    };
    if (ahead) {          // step 0: the recurrent halves from the initial state, up to three layers per launch
        for (int l0 = 0; l0 < L; l0 += 3) {
            SkinnyPlainJob pj[3];
            const int nj = std::min(3, L - l0);
            for (int i = 0; i < nj; ++i) pj[i] = rec_job(l0 + i, 0);
            MMQG_TRY(skinny_plain_multi(pj, nj, s));
        }
    }
    for (int t = 0; t < T; ++t) {
        float* sc = d.scores + (int64_t)t * B * ldS;
        float* at = d.attn + (int64_t)t * B * ldS;
        float* cx = d.ctx + (int64_t)t * B * C;
        // scores += h_top(t-1) * W_attn[:, E:]^T     (decoder.py:78,84,92: query = [emb | h_top]), softmax, contexts:
        // ONE launch when the caller gave the fused kernel its workspace and the operands allow it (attention_fused.hip)
        int fused_attn = 1;
        if (d.attn_ws && !g_no_fuse())
            fused_attn = attn_fused_fwd(v, sc, ldS, htop_base + t * BH, H, d.w_attn + E, Q, H, at, ldS, cx, C, d.attn_ws,
                                        d.attn_ws_bytes, s);
        if (fused_attn < 0) return fused_attn;
        const bool top_ahead = ahead && t > 0;        // gates_{L-1}[t] still lacks h_top(t-1) W_hh^T (step 0: done above)
        if (fused_attn != 0) {
            const SkinnyPair spr{htop_base + t * BH, H, d.w_attn + E, Q, H, 0};
            if (!g_no_fuse() && skinny_usable(&spr, 1)) {
                SkinnyPlainJob pj[2] = {};
                pj[0].M = B; pj[0].N = S; pj[0].npairs = 1; pj[0].pairs[0] = spr; pj[0].beta = 1; pj[0].C = sc; pj[0].ldc = ldS;
                if (top_ahead) pj[1] = rec_job(L - 1, t);
                MMQG_TRY(skinny_plain_multi(pj, top_ahead ? 2 : 1, s));
            } else {
                MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, B, S, H, htop_base + t * BH, H, d.w_attn + E, Q, nullptr, 0,
                                  nullptr, 0, 0, nullptr, nullptr, 1, sc, ldS, -1, s));
                if (top_ahead) { const SkinnyPlainJob j = rec_job(L - 1, t); MMQG_TRY(skinny_plain_multi(&j, 1, s)); }
            }
            MMQG_TRY(attn_softmax_context_fwd(v, sc, ldS, at, ldS, cx, C, s));
        } else if (top_ahead) {
            const SkinnyPlainJob j = rec_job(L - 1, t);
            MMQG_TRY(skinny_plain_multi(&j, 1, s));
        }
        for (int l = 0; l < L; ++l) {
            float* hs_l = d.hs + (int64_t)l * (T + 1) * BH;
            float* cs_l = d.cs + (int64_t)l * (T + 1) * BH;
            float* gates = d.gates + (int64_t)l * T * G + t * G;
            SkinnyPair prs[2];
            if (l == 0) {
                prs[0] = SkinnyPair{cx, C, d.w_ih[0] + E, In0, C, 0};
                prs[1] = SkinnyPair{hs_l + t * BH, H, d.w_hh[0], H, H, 0};
            } else {
                const float* xin = drop ? d.hdrop + (int64_t)(l - 1) * T * BH + t * BH
                                        : d.hs + (int64_t)(l - 1) * (T + 1) * BH + (t + 1) * BH;
                prs[0] = SkinnyPair{xin, H, d.w_ih[l], H, H, 0};
                prs[1] = SkinnyPair{hs_l + t * BH, H, d.w_hh[l], H, H, 0};
            }
            const bool fused = (H % 4 == 0) && skinny_usable(prs, 2) && !g_no_fuse();
            if (!fused) {
                if (l == 0)   // gates0 += ctx * W_ih0[:, E:]^T + h0 * W_hh0^T
                    MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, B, 4 * H, C, cx, C, d.w_ih[0] + E, In0, hs_l + t * BH, H,
                                      d.w_hh[0], H, H, nullptr, nullptr, 1, gates, 4 * H, -1, s));
                else
                    MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, B, 4 * H, H, prs[0].A, H, d.w_ih[l], H, hs_l + t * BH, H,
                                      d.w_hh[l], H, H, d.b_ih[l], d.b_hh[l], 0, gates, 4 * H, -1, s));
            }
            CellFwd c{};
            c.B = B; c.H = H; c.gates = gates; c.ld_g = 4 * H;
            c.h_prev = hs_l + t * BH; c.c_prev = cs_l + t * BH;
            c.h_out = hs_l + (t + 1) * BH; c.c_out = cs_l + (t + 1) * BH;
            c.h_drop = (drop && l < L - 1) ? d.hdrop + (int64_t)l * T * BH + t * BH : nullptr;
            c.lens = d.lens; c.t = t;
            c.p = drop ? d.dropout_p : 0.f; c.seed = d.seed; c.seed_off = d.seed_offset; c.stream_id = d.stream_base + (uint64_t)l * T + t;
            if (fused && ahead) {
                // only the late operand pair; the pre-activation slot already holds the recurrent half (+ for layer 0
                // the hoisted part).  Rides along: the recurrent half of the layer below for the NEXT step.
                SkinnyFwdJob job{};
                job.pairs[0] = prs[0]; job.npairs = 1; job.gates_has_pre = 1;
                job.bias1 = l == 0 ? nullptr : d.b_ih[l]; job.bias2 = l == 0 ? nullptr : d.b_hh[l];
                job.cell = c;
                SkinnyPlainJob look{};
                const bool with_look = l > 0 && t + 1 < T;
                if (with_look) look = rec_job(l - 1, t + 1);
                MMQG_TRY(skinny_cell_fwd_plus(job, &look, with_look ? 1 : 0, s));
            } else if (fused) {
                // layer 0: the hoisted emb part + biases already sit in gates; layers > 0 start from the biases
                if (l == 0) MMQG_TRY(skinny_cell_fwd(prs, 2, 1, nullptr, nullptr, c, s));
                else MMQG_TRY(skinny_cell_fwd(prs, 2, 0, d.b_ih[l], d.b_hh[l], c, s));
            } else {
                MMQG_TRY(lstm_cell_fwd(c, s));
            }
        }
    }
    return 0;
}

int decoder_seq_bwd(const mmqg_decoder_seq& d, const mmqg_decoder_seq_grad& g, hipStream_t s) {
    MMQG_TRY(check_decoder(d, "decoder_seq_bwd"));
    if (d.B == 0 || d.T == 0) return 0;
    MMQG_REQUIRE(g.dhtop && g.dgates && g.dscores && g.dctx && g.dh && g.dc && g.dxa, "decoder_seq_bwd: null buffer");
    const int T = d.T, B = d.B, H = d.H, L = d.L, E = d.E;
    const mmqg_attn_values& v = d.values;
    const int S = v.Lt + 2 * v.Lav, C = v.H + v.Da + v.Dv, Q = E + H, In0 = E + C;
    const int ldS = d.ld_attn, ldD = g.ld_ds;
    MMQG_REQUIRE(ldD >= S, "decoder_seq_bwd: ld_ds too small");
    const int64_t BH = (int64_t)B * H, G = (int64_t)B * 4 * H;
    const bool drop = d.training && d.dropout_p > 0.f && L > 1;
    MMQG_REQUIRE(g.phase >= 0 && g.phase <= 2, "decoder_seq_bwd: phase must be 0, 1 or 2");
    const WideWsScope wide_scope(g.wide_ws, g.wide_ws_bytes);
    const bool do_loop = g.phase != 2, do_wgrad = g.phase != 1;
    if (do_loop) {
    // the whole time loop incl. the initial-state gradients as ONE persistent launch (persist_dec_bwd.hip) where the shape
    // is taken; the launched loop below otherwise
    bool loop_done = false;
    if (g.persist_ws && !g_no_fuse() && d.w_ih0cT && d.w_attn_hT) {
        const int rc = decoder_seq_bwd_persistent(d, g, s);
        if (rc < 0) return rc;
        loop_done = rc == 0;
    }
    if (!loop_done) {
    {
        const CopySeg z[2] = {CopySeg{g.dh, nullptr, (int64_t)L * BH}, CopySeg{g.dc, nullptr, (int64_t)L * BH}};
        MMQG_TRY(copy_or_zero_multi(z, 2, s));
    }
    // fused backward: possible when k-major (transposed) copies of every recurrent weight are given
    bool fusedb = !g_no_fuse() && d.w_ih0cT && d.w_attn_hT && (ldS % 4 == 0) && (ldD % 4 == 0);
    for (int l = 0; l < L && fusedb; ++l) {
        const SkinnyPair a{g.dgates, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
        fusedb = d.w_hhT[l] && skinny_usable(&a, 1);
        if (fusedb && l > 0) {
            const SkinnyPair b{g.dgates, 4 * H, d.w_ihT[l], 4 * H, 4 * H, 1};
            fusedb = d.w_ihT[l] && skinny_usable(&b, 1);
        }
    }
    if (fusedb) {
        const SkinnyPair a{g.dgates, 4 * H, d.w_ih0cT, 4 * H, 4 * H, 0};
        const SkinnyPair b{g.dscores, ldD, d.w_attn_hT, ldS, ldS, 0};
        fusedb = skinny_usable(&a, 1) && skinny_usable(&b, 1) && (ldD >= ldS);
    }
    // Look-ahead: the recurrent product dgates_l(t) * W_hh_l is needed by cell (l, t-1), a whole step
    // after dgates_l(t) exists.  It is formed as an extra (plain) job of the NEXT launch on the dependent
    // chain — cell (l-1, t), or the dctx product for layer 0 — into dh_pre[l], so the cell kernels on the
    // chain only carry the operand pair that really is late (K halves for layers 0..L-2, 2048 -> 488 for the
    // top layer) and those launches fill twice as many CUs.
    const bool ahead = fusedb && g.dh_pre && T > 1;
    for (int t = T - 1; t >= 0; --t) {
        for (int l = L - 1; l >= 0; --l) {
            const float* cs_l = d.cs + (int64_t)l * (T + 1) * BH;
            float* dg = g.dgates + (int64_t)l * T * G + t * G;
            CellBwd c{};
            c.B = B; c.H = H; c.gates_act = d.gates + (int64_t)l * T * G + t * G;
            c.c_prev = cs_l + t * BH; c.c_new = cs_l + (t + 1) * BH;
            c.dh_rec = g.dh + l * BH;
            if (l < L - 1) {
                c.p = drop ? d.dropout_p : 0.f; c.seed = d.seed; c.seed_off = d.seed_offset; c.stream_id = d.stream_base + (uint64_t)l * T + t;
                if (!fusedb) { c.dh_above = g.dxa + l * BH; c.above_stride_b = H; }
            } else {
                c.dh_extra = g.dhtop + t * BH; c.extra_stride_b = H;
            }
            c.dc = g.dc + l * BH; c.dgates = dg; c.ld_dg = 4 * H;
            c.lens = d.lens; c.t = t;
            if (fusedb) {
                // dh_l(t) = dgates_l(t+1) W_hh_l  [+ dscores(t+1) W_attn_h for the top layer: the query of step
                // t+1 used h_top(t)]  [+ mask_l(t) * dgates_{l+1}(t) W_ih_{l+1}: gradient of the input of the
                // layer above]  + carried / extra terms; then the cell backward — one launch.
                SkinnyPair prs[3];
                int np = 0;
                if (t < T - 1) {
                    if (ahead) c.dh_pre = g.dh_pre + l * BH;       // formed one launch after cell (l, t+1)
                    else prs[np++] = SkinnyPair{g.dgates + (int64_t)l * T * G + (t + 1) * G, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
                    if (l == L - 1)
                        prs[np++] = SkinnyPair{g.dscores + (int64_t)(t + 1) * B * ldD, ldD, d.w_attn_hT, ldS, ldS, 0};
                }
                if (l < L - 1)
                    prs[np++] = SkinnyPair{g.dgates + (int64_t)(l + 1) * T * G + t * G, 4 * H, d.w_ihT[l + 1], 4 * H, 4 * H, 1};
                // product for cell (l+1, t-1): dgates_{l+1}(t) exists since the previous launch
                SkinnyPlainJob look{};
                const bool with_look = ahead && t > 0 && l < L - 1;
                if (with_look) {
                    look.M = B; look.N = H; look.npairs = 1; look.C = g.dh_pre + (l + 1) * BH; look.ldc = H;
                    look.pairs[0] = SkinnyPair{g.dgates + (int64_t)(l + 1) * T * G + t * G, 4 * H, d.w_hhT[l + 1], 4 * H, 4 * H, 0};
                }
                if (np > 0) {
                    SkinnyBwdJob job{};
                    for (int i = 0; i < np; ++i) job.pairs[i] = prs[i];
                    job.npairs = np; job.cell = c;
                    MMQG_TRY(skinny_cell_bwd_plus(job, &look, with_look ? 1 : 0, s));
                } else {
                    MMQG_TRY(lstm_cell_bwd(c, s));
                    if (with_look) MMQG_TRY(skinny_plain_multi(&look, 1, s));
                }
                if (l == 0) {   // dctx(t) = dgates_0(t) * W_ih0[:, E:]  (+ layer 0's look-ahead product)
                    SkinnyPlainJob pj[2] = {};
                    pj[0].M = B; pj[0].N = C; pj[0].npairs = 1; pj[0].C = g.dctx + (int64_t)t * B * C; pj[0].ldc = C;
                    pj[0].pairs[0] = SkinnyPair{dg, 4 * H, d.w_ih0cT, 4 * H, 4 * H, 0};
                    int nj = 1;
                    if (ahead && t > 0) {
                        pj[1].M = B; pj[1].N = H; pj[1].npairs = 1; pj[1].C = g.dh_pre; pj[1].ldc = H;
                        pj[1].pairs[0] = SkinnyPair{dg, 4 * H, d.w_hhT[0], 4 * H, 4 * H, 0};
                        nj = 2;
                    }
                    MMQG_TRY(skinny_plain_multi(pj, nj, s));
                }
                continue;
            }
            MMQG_TRY(lstm_cell_bwd(c, s));
            MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, B, H, 4 * H, dg, 4 * H, d.w_hh[l], H, nullptr, 0, nullptr, 0, 0,
                              nullptr, nullptr, 1, g.dh + l * BH, H, -1, s));
            if (l > 0) {
                MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, B, H, 4 * H, dg, 4 * H, d.w_ih[l], H, nullptr, 0, nullptr,
                                  0, 0, nullptr, nullptr, 0, g.dxa + (l - 1) * BH, H, -1, s));
            } else {
                MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, B, C, 4 * H, dg, 4 * H, d.w_ih[0] + E, In0, nullptr, 0,
                                  nullptr, 0, 0, nullptr, nullptr, 0, g.dctx + (int64_t)t * B * C, C, -1, s));
            }
        }
        float* ds = g.dscores + (int64_t)t * B * ldD;
        // (one kernel: the softmax Jacobian's row dot is ctx(t) . dctx(t), the forward's saved context)
        MMQG_TRY(attn_context_bwd_fused(v, d.attn + (int64_t)t * B * ldS, ldS, d.ctx + (int64_t)t * B * C, C,
                                        g.dctx + (int64_t)t * B * C, C, ds, ldD, s));
        if (!fusedb)   // gradient of the query's h_top(t-1) half: feeds the top layer's recurrent gradient
            MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, B, H, S, ds, ldD, d.w_attn + E, Q, nullptr, 0, nullptr, 0, 0,
                              nullptr, nullptr, 1, g.dh + (int64_t)(L - 1) * BH, H, -1, s));
    }
    if (fusedb) {
        // gradient of the initial state (the text encoder's final state): carry + dgates_l(0) W_hh_l
        // (+ dscores(0) W_attn_h for the top layer, whose h0 was the query of step 0)
        for (int l0 = 0; l0 < L; l0 += 3) {     // up to three layers per launch
            SkinnyPlainJob pj[3] = {};
            const int nj = std::min(3, L - l0);
            for (int i = 0; i < nj; ++i) {
                const int l = l0 + i;
                pj[i].M = B; pj[i].N = H; pj[i].beta = 1; pj[i].C = g.dh + l * BH; pj[i].ldc = H;
                pj[i].pairs[pj[i].npairs++] = SkinnyPair{g.dgates + (int64_t)l * T * G, 4 * H, d.w_hhT[l], 4 * H, 4 * H, 0};
                if (l == L - 1) pj[i].pairs[pj[i].npairs++] = SkinnyPair{g.dscores, ldD, d.w_attn_hT, ldS, ldS, 0};
            }
            MMQG_TRY(skinny_plain_multi(pj, nj, s));
        }
    }
    }   // !loop_done
    // gradient of the value rows an encoder produced (text rows feed the text encoder's
    // backward, video rows the frame encoder's); audio features are inputs and get none
    if (g.dtext && g.n_text_rows > 0)
        MMQG_TRY(attn_dvalues(T, B, std::min(g.n_text_rows, v.Lt), v.H, d.attn, (int64_t)B * ldS, ldS, 0, g.dctx,
                              (int64_t)B * C, C, 0, g.dtext, g.dtext_stride_row, g.dtext_stride_b, 0, s));
    if (g.dvideo && g.n_video_rows > 0)
        MMQG_TRY(attn_dvalues(T, B, std::min(g.n_video_rows, v.Lav), v.Dv, d.attn, (int64_t)B * ldS, ldS, v.Lt + v.Lav,
                              g.dctx, (int64_t)B * C, C, v.H + v.Da, g.dvideo, g.dvideo_stride_row, g.dvideo_stride_b, 0, s));
    }   // do_loop
    if (!do_wgrad) return 0;
    const int R = T * B;
    const float* htop_prev = d.hs + (int64_t)(L - 1) * (T + 1) * BH;   // rows t = h_top(t-1)
    // embedded-word gradient: layer-0 gate path + score path
    if (g.dxemb) {
        MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, R, E, 4 * H, g.dgates, 4 * H, d.w_ih[0], In0, nullptr, 0, nullptr,
                          0, 0, nullptr, nullptr, 0, g.dxemb, E, -1, s));
        MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_MN_MAJOR, R, E, S, g.dscores, ldD, d.w_attn, Q, nullptr, 0, nullptr, 0, 0,
                          nullptr, nullptr, 1, g.dxemb, E, -1, s));
    }
    // every weight gradient of the decoder (dW += dX^T Y over all T*B rows) as grouped launches
    GemmProblem wg[2 * MMQG_MAX_LAYERS + 3];
    float* cs1[2 * MMQG_MAX_LAYERS + 3] = {};
    float* cs2[2 * MMQG_MAX_LAYERS + 3] = {};
    int nw = 0;
    if (g.dw_attn) {
        wg[nw++] = GemmProblem{S, E, R, g.dscores, ldD, d.xemb, E, g.dw_attn, Q, 1};
        wg[nw++] = GemmProblem{S, H, R, g.dscores, ldD, htop_prev, H, g.dw_attn + E, Q, 1};
        if (g.db_attn) cs1[nw - 1] = g.db_attn;          // the score bias gradient rides in this product's staging pass
    } else if (g.db_attn) {
        MMQG_TRY(colsum_add(g.dscores, ldD, R, S, g.db_attn, s));
    }
    for (int l = 0; l < L; ++l) {
        const float* dg_l = g.dgates + (int64_t)l * T * G;
        const float* hs_l = d.hs + (int64_t)l * (T + 1) * BH;
        if (g.dw_ih[l]) {
            if (l == 0) {
                wg[nw++] = GemmProblem{4 * H, E, R, dg_l, 4 * H, d.xemb, E, g.dw_ih[0], In0, 1};
                wg[nw++] = GemmProblem{4 * H, C, R, dg_l, 4 * H, d.ctx, C, g.dw_ih[0] + E, In0, 1};
            } else {
                const float* X = drop ? d.hdrop + (int64_t)(l - 1) * T * BH : d.hs + (int64_t)(l - 1) * (T + 1) * BH + BH;
                wg[nw++] = GemmProblem{4 * H, H, R, dg_l, 4 * H, X, H, g.dw_ih[l], H, 1};
            }
        }
        const int before = nw;
        if (g.dw_hh[l]) wg[nw++] = GemmProblem{4 * H, H, R, dg_l, 4 * H, hs_l, H, g.dw_hh[l], H, 1};
        float* b1 = g.db_ih[l] ? g.db_ih[l] : g.db_hh[l];
        float* b2 = g.db_ih[l] ? g.db_hh[l] : nullptr;
        if (b1 && nw > before) { cs1[nw - 1] = b1; cs2[nw - 1] = b2; }        // bias gradients ride in the dW_hh product
        else if (b1) MMQG_TRY(colsum_add2(dg_l, 4 * H, R, 4 * H, b1, b2, s));
    }
    MMQG_TRY(gemm_f32_wgrad_group(wg, cs1, cs2, nw, s));
    return 0;
}

}  // namespace mmqg

// ------------------------------------------------------------------------- free-running decode
namespace mmqg {

// C = (beta?C:0) + bias1 + bias2 + sum of pairs; the fused skinny kernel when the operands allow it,
// the tiled GEMM otherwise
static int pairs_product(int M, int N, const SkinnyPair* pr, int np, const float* bias1, const float* bias2, int beta,
                         float* C, int ldc, hipStream_t s) {
    if (!g_no_fuse() && !bias2 && skinny_usable(pr, np)) return skinny_plain(M, N, pr, np, bias1, beta, C, ldc, s);
    for (int i = 0; i < np; ++i)
        MMQG_TRY(gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, M, N, pr[i].K, pr[i].A, pr[i].lda, pr[i].B, pr[i].ldb, nullptr, 0,
                          nullptr, 0, 0, i == 0 ? bias1 : nullptr, i == 0 ? bias2 : nullptr, i == 0 ? beta : 1, C, ldc, 1, s));
    return 0;
}

int decoder_decode(const mmqg_decoder_decode& d, hipStream_t s) {
    MMQG_REQUIRE(d.T >= 0 && d.B >= 0 && d.H > 0 && d.E > 0 && d.V > 0, "decoder_decode: bad shape");
    MMQG_REQUIRE(d.L >= 1 && d.L <= MMQG_MAX_LAYERS, "decoder_decode: L must be in [1,%d]", MMQG_MAX_LAYERS);
    MMQG_REQUIRE(d.values.B == d.B, "decoder_decode: values.B must equal B");
    MMQG_REQUIRE(d.emb_table && d.w_attn && d.b_attn && d.w_out && d.b_out && d.h0 && d.c0, "decoder_decode: null input");
    MMQG_REQUIRE(d.ids && d.attn && d.xemb && d.scores && d.ctx && d.gates && d.hs && d.cs && d.logits,
                 "decoder_decode: null buffer");
    MMQG_REQUIRE(d.strategy == 0 || d.strategy == 1, "decoder_decode: strategy must be 0 (greedy) or 1 (sampling)");
    MMQG_REQUIRE(!d.loss_rows || d.target, "decoder_decode: loss requested without targets");
    for (int l = 0; l < d.L; ++l)
        MMQG_REQUIRE(d.w_ih[l] && d.w_hh[l] && d.b_ih[l] && d.b_hh[l], "decoder_decode: null weight (layer %d)", l);
    if (d.B == 0) return 0;
    const int T = d.T, B = d.B, H = d.H, L = d.L, E = d.E, V = d.V;
    const mmqg_attn_values& v = d.values;
    const int S = v.Lt + 2 * v.Lav, C = v.H + v.Da + v.Dv, Q = E + H, In0 = E + C, ldS = d.ld_attn;
    MMQG_REQUIRE(ldS >= S, "decoder_decode: ld_attn too small");
    const int64_t BH = (int64_t)B * H, LBH = (int64_t)L * BH, G = (int64_t)B * 4 * H;
    MMQG_TRY(copy_or_zero(d.hs, d.h0, (size_t)LBH, s));
    MMQG_TRY(copy_or_zero(d.cs, d.c0, (size_t)LBH, s));
    MMQG_TRY(fill_i64(d.ids, d.start_id, B, s));
    for (int t = 0; t < T; ++t) {
        const float* h_in = d.hs + (t & 1) * LBH;
        const float* c_in = d.cs + (t & 1) * LBH;
        float* h_out = d.hs + ((t + 1) & 1) * LBH;
        float* c_out = d.cs + ((t + 1) & 1) * LBH;
        float* at = d.attn + (int64_t)t * B * ldS;
        MMQG_TRY(embedding_fwd(d.emb_table, d.ids + (int64_t)t * B, d.xemb, B, V, E, E, s));            // decoder.py:75
        // scores = [emb | h_top] * W_attn^T + b                                                         decoder.py:78,84,92
        SkinnyPair sp[2] = {{d.xemb, E, d.w_attn, Q, E, 0}, {h_in + (int64_t)(L - 1) * BH, H, d.w_attn + E, Q, H, 0}};
        MMQG_TRY(pairs_product(B, S, sp, 2, d.b_attn, nullptr, 0, d.scores, ldS, s));
        MMQG_TRY(attn_softmax_context_fwd(v, d.scores, ldS, at, ldS, d.ctx, C, s));
        for (int l = 0; l < L; ++l) {
            float* gates = d.gates + (int64_t)l * G;
            SkinnyPair pr[3];
            int np = 0;
            if (l == 0) {
                pr[np++] = SkinnyPair{d.xemb, E, d.w_ih[0], In0, E, 0};
                pr[np++] = SkinnyPair{d.ctx, C, d.w_ih[0] + E, In0, C, 0};
            } else {
                pr[np++] = SkinnyPair{h_out + (int64_t)(l - 1) * BH, H, d.w_ih[l], H, H, 0};
            }
            pr[np++] = SkinnyPair{h_in + l * BH, H, d.w_hh[l], H, H, 0};
            CellFwd c{};
            c.B = B; c.H = H; c.gates = gates; c.ld_g = 4 * H;
            c.h_prev = h_in + l * BH; c.c_prev = c_in + l * BH; c.h_out = h_out + l * BH; c.c_out = c_out + l * BH;
            if (!g_no_fuse() && H % 4 == 0 && skinny_usable(pr, np)) {
                MMQG_TRY(skinny_cell_fwd(pr, np, 0, d.b_ih[l], d.b_hh[l], c, s));
            } else {
                MMQG_TRY(pairs_product(B, 4 * H, pr, np, d.b_ih[l], d.b_hh[l], 0, gates, 4 * H, s));
                MMQG_TRY(lstm_cell_fwd(c, s));
            }
        }
        float* logits = d.keep_logits ? d.logits + (int64_t)t * B * V : d.logits;
        const SkinnyPair op{h_out + (int64_t)(L - 1) * BH, H, d.w_out, H, H, 0};
        MMQG_TRY(pairs_product(B, V, &op, 1, d.b_out, nullptr, 0, logits, V, s));                        // decoder.py:106
        int64_t* next = d.ids + (int64_t)(t + 1) * B;
        MMQG_TRY(ce_fwd_bwd(logits, V, d.target ? d.target + (int64_t)t * B : nullptr,
                            d.row_weight ? d.row_weight + (int64_t)t * B : nullptr, B, V,
                            d.loss_rows ? d.loss_rows + (int64_t)t * B : nullptr, next, nullptr, 0, s));
        if (d.strategy == 1) MMQG_TRY(sample_gumbel(logits, V, B, V, d.seed, (uint64_t)t, next, s));
    }
    return 0;
}

}  // namespace mmqg
