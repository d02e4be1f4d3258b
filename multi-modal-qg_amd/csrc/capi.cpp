// extern "C" surface of libmmqg_hip.so: thin, checked forwarding to the launchers.
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace mmqg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

int lstm_seq_fwd(const mmqg_lstm_seq& d, hipStream_t s);
int lstm_seq_bwd(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, hipStream_t s);
int lstm_seq_bwd_pair(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, const mmqg_lstm_seq& d2, const mmqg_lstm_seq_grad& g2,
                      hipStream_t s);
int64_t lstm_persist_ws_bytes(int T, int B, int L, int H);
int persist_launch_count();
void persist_set_trace(unsigned long long* buf, int64_t words);
int decoder_seq_fwd(const mmqg_decoder_seq& d, hipStream_t s);
int decoder_seq_bwd(const mmqg_decoder_seq& d, const mmqg_decoder_seq_grad& g, hipStream_t s);
int decoder_decode(const mmqg_decoder_decode& d, hipStream_t s);

}  // namespace mmqg

using namespace mmqg;
static inline hipStream_t S(mmqg_stream s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

int mmqg_abi_version(void) { return MMQG_ABI_VERSION; }
const char* mmqg_last_error(void) { return get_error(); }

int mmqg_gemm_f32(int a_layout, int b_layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                  const float* A2, int lda2, const float* B2, int ldb2, int K2, const float* bias, const float* bias2,
                  int beta, float* C, int ldc, int split_k, mmqg_stream stream) {
    return gemm_f32(a_layout, b_layout, M, N, K, A, lda, B, ldb, A2, lda2, B2, ldb2, K2, bias, bias2, beta, C, ldc,
                    split_k, S(stream));
}

int mmqg_gemm_f32_grouped(int a_layout, int b_layout, const mmqg_gemm_problem* problems, int n, mmqg_stream stream) {
    MMQG_REQUIRE(a_layout == MMQG_K_MAJOR || a_layout == MMQG_MN_MAJOR, "gemm_f32_grouped: bad a_layout");
    MMQG_REQUIRE(b_layout == MMQG_K_MAJOR || b_layout == MMQG_MN_MAJOR, "gemm_f32_grouped: bad b_layout");
    return gemm_f32_grouped(a_layout, b_layout, problems, n, S(stream));
}
int mmqg_pack_batch(const mmqg_batch_pack* a, mmqg_stream stream) {
    MMQG_REQUIRE(a, "mmqg_pack_batch: null descriptor");
    return pack_batch(*a, S(stream));
}
int mmqg_fetch_mapped(const mmqg_copy_seg* segs, int n, mmqg_stream stream) { return fetch_mapped(segs, n, S(stream)); }
int mmqg_embedding_fwd(const float* table, const int64_t* ids, float* out, int n, int V, int E, int ld_out,
                       mmqg_stream stream) {
    return embedding_fwd(table, ids, out, n, V, E, ld_out, S(stream));
}
int mmqg_embedding_bwd(const float* dout, int ld_dout, const int64_t* ids, float* dtable, int n, int V, int E,
                       mmqg_stream stream) {
    return embedding_bwd(dout, ld_dout, ids, dtable, n, V, E, S(stream));
}

int mmqg_attn_softmax_context_fwd(const mmqg_attn_values* v, const float* scores, int ld_s, float* attn, int ld_a,
                                  float* ctx, int ld_c, mmqg_stream stream) {
    MMQG_REQUIRE(v, "mmqg_attn_softmax_context_fwd: null descriptor");
    return attn_softmax_context_fwd(*v, scores, ld_s, attn, ld_a, ctx, ld_c, S(stream));
}
int64_t mmqg_attn_fused_ws_bytes(const mmqg_attn_values* v, int Hq) { return v ? attn_fused_ws_bytes(*v, Hq) : 0; }
int mmqg_attn_scores_softmax_context_fwd(const mmqg_attn_values* v, const float* pre, int ld_s, const float* h, int ld_h,
                                         const float* W, int ld_w, int Hq, float* attn, int ld_a, float* ctx, int ld_c,
                                         float* ws, int64_t ws_bytes, mmqg_stream stream) {
    MMQG_REQUIRE(v, "mmqg_attn_scores_softmax_context_fwd: null descriptor");
    MMQG_REQUIRE(pre && h && W && attn && ctx, "mmqg_attn_scores_softmax_context_fwd: null pointer");
    return attn_fused_fwd(*v, pre, ld_s, h, ld_h, W, ld_w, Hq, attn, ld_a, ctx, ld_c, ws, ws_bytes, S(stream));
}
int mmqg_attn_context_bwd(const mmqg_attn_values* v, const float* attn, int ld_a, const float* dctx, int ld_c,
                          const float* dattn, int ld_da, float* dscores, int ld_ds, mmqg_stream stream) {
    MMQG_REQUIRE(v, "mmqg_attn_context_bwd: null descriptor");
    return attn_context_bwd(*v, attn, ld_a, dctx, ld_c, dattn, ld_da, dscores, ld_ds, S(stream));
}
int mmqg_attn_context_bwd_fused(const mmqg_attn_values* v, const float* attn, int ld_a, const float* ctx, int ld_x,
                                const float* dctx, int ld_c, float* dscores, int ld_ds, mmqg_stream stream) {
    MMQG_REQUIRE(v, "mmqg_attn_context_bwd_fused: null descriptor");
    return attn_context_bwd_fused(*v, attn, ld_a, ctx, ld_x, dctx, ld_c, dscores, ld_ds, S(stream));
}
int mmqg_attn_dvalues(int T, int B, int n_rows, int D, const float* attn, int64_t attn_stride_t, int ld_a, int seg_off,
                      const float* dctx, int64_t dctx_stride_t, int ld_c, int ctx_off, float* out,
                      int64_t out_stride_row, int64_t out_stride_b, int accumulate, mmqg_stream stream) {
    return attn_dvalues(T, B, n_rows, D, attn, attn_stride_t, ld_a, seg_off, dctx, dctx_stride_t, ld_c, ctx_off, out,
                        out_stride_row, out_stride_b, accumulate, S(stream));
}

int mmqg_lstm_cell_fwd(int B, int H, float* gates, int ld_g, const float* h_prev, const float* c_prev, float* h_out,
                       float* c_out, float* h_drop, float* y_out, int64_t y_stride_b, const int32_t* lens, int t,
                       float dropout_p, uint64_t seed, uint64_t stream_id, mmqg_stream stream) {
    CellFwd c{};
    c.B = B; c.H = H; c.gates = gates; c.ld_g = ld_g; c.h_prev = h_prev; c.c_prev = c_prev;
    c.h_out = h_out; c.c_out = c_out; c.h_drop = h_drop; c.y_out = y_out; c.y_stride_b = y_stride_b;
    c.lens = lens; c.t = t; c.p = dropout_p; c.seed = seed; c.stream_id = stream_id;
    return lstm_cell_fwd(c, S(stream));
}
int mmqg_lstm_cell_bwd(int B, int H, const float* gates_act, const float* c_prev, const float* c_new, float* dh_rec,
                       const float* dh_above, int64_t above_stride_b, float dropout_p, uint64_t seed,
                       uint64_t stream_id, const float* dh_extra, int64_t extra_stride_b, float* dc, float* dgates,
                       int ld_dg, const int32_t* lens, int t, mmqg_stream stream) {
    CellBwd c{};
    c.B = B; c.H = H; c.gates_act = gates_act; c.c_prev = c_prev; c.c_new = c_new; c.dh_rec = dh_rec;
    c.dh_above = dh_above; c.above_stride_b = above_stride_b; c.p = dropout_p; c.seed = seed; c.stream_id = stream_id;
    c.dh_extra = dh_extra; c.extra_stride_b = extra_stride_b; c.dc = dc; c.dgates = dgates; c.ld_dg = ld_dg;
    c.lens = lens; c.t = t;
    return lstm_cell_bwd(c, S(stream));
}
int mmqg_dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t stream_id, const int32_t* seed_offset,
                      mmqg_stream stream) {
    return dropout_mask(out, n, p, seed, stream_id, seed_offset, S(stream));
}

int mmqg_ce_fwd_bwd(const float* logits, int ld, const int64_t* target, const float* row_weight, int rows, int V,
                    float* loss_rows, int64_t* argmax, float* dlogits, int ld_d, mmqg_stream stream) {
    return ce_fwd_bwd(logits, ld, target, row_weight, rows, V, loss_rows, argmax, dlogits, ld_d, S(stream));
}
int mmqg_ce_fwd_bwd_stats(const float* logits, int ld, const int64_t* target, const float* row_weight, int rows, int V,
                          const float* stats, int stats_tiles, float* loss_rows, int64_t* argmax, float* dlogits, int ld_d,
                          mmqg_stream stream) {
    MMQG_REQUIRE(stats_tiles >= 0 && (stats_tiles == 0 || stats), "ce_fwd_bwd_stats: stats_tiles > 0 needs a stats buffer");
    return ce_fwd_bwd(logits, ld, target, row_weight, rows, V, loss_rows, argmax, dlogits, ld_d, S(stream), stats,
                      stats_tiles);
}
int mmqg_linear_wgrad(int out_features, int in_features, int rows, const float* dY, int ld_dy, const float* X, int ldx,
                      float* dW, int lddw, float* dbias, mmqg_stream stream) {
    if (out_features <= 0 || in_features <= 0 || rows <= 0) return 0;
    MMQG_REQUIRE(dY && X && dW, "linear_wgrad: null operand");
    MMQG_REQUIRE(ld_dy >= out_features && ldx >= in_features && lddw >= in_features, "linear_wgrad: leading dimension too small");
    const GemmProblem q{out_features, in_features, rows, dY, ld_dy, X, ldx, dW, lddw, 1};
    float* cs[1] = {dbias};
    return gemm_f32_wgrad_group(&q, dbias ? cs : nullptr, nullptr, 1, S(stream));
}
static int g_projection_kernel = 0;
int mmqg_projection_last_kernel(void) { return g_projection_kernel; }
int64_t mmqg_projection_stats_ws_bytes(int rows, int V) { return rows > 0 && V > 0 ? gemm_nt_stats_bytes(rows, V) : 0; }
int mmqg_projection_fwd(int rows, int V, int H, const float* h, int ldh, const float* W, int ldw, const float* bias,
                        float* logits, int ld, float* stats, int64_t stats_bytes, int32_t* stats_tiles,
                        mmqg_stream stream) {
    MMQG_REQUIRE(stats_tiles, "projection_fwd: stats_tiles must point to an int");
    *stats_tiles = 0;
    if (rows <= 0 || V <= 0) return 0;
    MMQG_REQUIRE(H > 0 && h && W && logits, "projection_fwd: null operand");
    MMQG_REQUIRE(ldh >= H && ldw >= H && ld >= V, "projection_fwd: leading dimension too small");
    // the split-bf16 kernel first (fp32-exact operands on the bf16 matrix cores), then the fp32 one-tile-per-CU kernel
    int rc = gemm_x3_projection(rows, V, H, h, ldh, W, ldw, bias, logits, ld, stats, stats_bytes, stats_tiles, S(stream));
    if (rc <= 0) { g_projection_kernel = 2; return rc; }
    rc = gemm_nt_tile(rows, V, H, h, ldh, W, ldw, bias, nullptr, logits, ld, S(stream), stats, stats_bytes, stats_tiles);
    if (rc <= 0) { g_projection_kernel = 1; return rc; }
    g_projection_kernel = 0;
    *stats_tiles = 0;
    return gemm_f32(MMQG_K_MAJOR, MMQG_K_MAJOR, rows, V, H, h, ldh, W, ldw, nullptr, 0, nullptr, 0, 0, bias, nullptr, 0,
                    logits, ld, 1, S(stream));
}
int mmqg_colsum_add(const float* X, int ld, int M, int N, float* out, mmqg_stream stream) {
    return colsum_add(X, ld, M, N, out, S(stream));
}
int mmqg_reduce_sum(const float* x, int n, float* out, mmqg_stream stream) { return reduce_sum(x, n, out, S(stream)); }

int mmqg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps,
                   const int32_t* step, float grad_scale, mmqg_stream stream) {
    return adam_step(p, g, m, v, n, lr, b1, b2, eps, step, grad_scale, S(stream));
}
int mmqg_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps,
                           const int32_t* step, float grad_scale, const int32_t* skip, mmqg_stream stream) {
    return adam_step(p, g, m, v, n, lr, b1, b2, eps, step, grad_scale, S(stream), skip);
}
int mmqg_persist_guard_refresh(int32_t* flag, mmqg_stream stream) { return persist_guard_refresh(flag, S(stream)); }
int mmqg_counter_add(int32_t* counter, int delta, mmqg_stream stream) { return counter_add(counter, delta, S(stream)); }
int mmqg_transpose_f32(const float* src, int ld_src, int rows, int cols, float* dst, int ld_dst, mmqg_stream stream) {
    return transpose_f32(src, ld_src, rows, cols, dst, ld_dst, S(stream));
}

int mmqg_transpose_f32_batch(const mmqg_transpose_job* jobs, int n, mmqg_stream stream) {
    return transpose_f32_batch(jobs, n, S(stream));
}

int mmqg_lstm_seq_fwd(const mmqg_lstm_seq* d, mmqg_stream stream) {
    MMQG_REQUIRE(d, "mmqg_lstm_seq_fwd: null descriptor");
    MMQG_TRY(persist_check_healthy("mmqg_lstm_seq_fwd"));
    return lstm_seq_fwd(*d, S(stream));
}
int64_t mmqg_lstm_seq_persist_ws_bytes(int T, int B, int L, int H) { return lstm_persist_ws_bytes(T, B, L, H); }
int mmqg_persist_launch_count(void) { return persist_launch_count(); }
int64_t mmqg_lstm_seq_bwd_persist_ws_bytes(int T, int B, int L, int H) { return lstm_persist_bwd_ws_bytes(T, B, L, H); }
int mmqg_persist_bwd_launch_count(void) { return persist_bwd_launch_count(); }
int64_t mmqg_decoder_seq_persist_ws_bytes(const mmqg_decoder_seq* d) { return d ? decoder_persist_ws_bytes(*d) : 0; }
int mmqg_decoder_persist_launch_count(void) { return decoder_persist_launch_count(); }
int64_t mmqg_decoder_seq_bwd_persist_ws_bytes(const mmqg_decoder_seq* d, const mmqg_decoder_seq_grad* g) {
    return (d && g) ? decoder_persist_bwd_ws_bytes(*d, g->ld_ds) : 0;
}
int mmqg_decoder_persist_bwd_launch_count(void) { return decoder_persist_bwd_launch_count(); }
int mmqg_decoder_persist_bwd_set_trace(uint64_t* buf, int64_t words) { decoder_persist_bwd_set_trace(reinterpret_cast<unsigned long long*>(buf), words); return 0; }
int mmqg_decoder_persist_set_trace(uint64_t* buf, int64_t words) { decoder_persist_set_trace(reinterpret_cast<unsigned long long*>(buf), words); return 0; }
int mmqg_persist_bwd_set_trace(uint64_t* buf, int64_t words) { persist_bwd_set_trace(reinterpret_cast<unsigned long long*>(buf), words); return 0; }
int64_t mmqg_wide_ws_bytes(int B, int max_N) { return skinny_wide_ws_bytes(B, max_N); }
int mmqg_persist_declined_count(void) { return persist_declined_count(); }
int mmqg_persist_failures(void) { return persist_failures(); }
int mmqg_persist_clear_failures(void) { persist_clear_failures(); return 0; }
int mmqg_persist_set_test_fault(int extra_workgroups, uint32_t max_spins) {
    MMQG_REQUIRE(extra_workgroups >= 0, "persist_set_test_fault: extra_workgroups must be >= 0");
    static const bool hooks = [] { const char* e = getenv("MMQG_ENABLE_TEST_HOOKS"); return e && atoi(e) != 0; }();
    MMQG_REQUIRE(hooks || (extra_workgroups == 0 && max_spins == 0),
                 "persist_set_test_fault: fault injection is only armed in a process started with MMQG_ENABLE_TEST_HOOKS=1");
    persist_set_test_fault(extra_workgroups, max_spins);
    return 0;
}
int mmqg_persist_set_reserved_cus(int n) {
    MMQG_REQUIRE(n >= 0, "persist_set_reserved_cus: n must be >= 0");
    persist_set_reserved_cus(n);
    return 0;
}
int mmqg_persist_usable_cus(mmqg_stream stream, int shrinkable) { return persist_usable_cus(S(stream), shrinkable != 0); }
int mmqg_persist_set_trace(uint64_t* buf, int64_t words) { persist_set_trace(reinterpret_cast<unsigned long long*>(buf), words); return 0; }
int mmqg_lstm_seq_bwd(const mmqg_lstm_seq* d, const mmqg_lstm_seq_grad* g, mmqg_stream stream) {
    MMQG_REQUIRE(d && g, "mmqg_lstm_seq_bwd: null descriptor");
    MMQG_TRY(persist_check_healthy("mmqg_lstm_seq_bwd"));
    return lstm_seq_bwd(*d, *g, S(stream));
}
int mmqg_lstm_seq_bwd_pair(const mmqg_lstm_seq* d, const mmqg_lstm_seq_grad* g, const mmqg_lstm_seq* d2,
                           const mmqg_lstm_seq_grad* g2, mmqg_stream stream) {
    MMQG_REQUIRE(d && g && d2 && g2, "mmqg_lstm_seq_bwd_pair: null descriptor");
    MMQG_TRY(persist_check_healthy("mmqg_lstm_seq_bwd_pair"));
    return lstm_seq_bwd_pair(*d, *g, *d2, *g2, S(stream));
}
int mmqg_frame_cnn_fwd(const mmqg_frame_cnn* d, mmqg_stream stream) {
    MMQG_REQUIRE(d, "mmqg_frame_cnn_fwd: null descriptor");
    return frame_cnn_fwd(*d, S(stream));
}
int mmqg_frame_cnn_bwd(const mmqg_frame_cnn* d, const mmqg_frame_cnn_grad* g, mmqg_stream stream) {
    MMQG_REQUIRE(d && g, "mmqg_frame_cnn_bwd: null descriptor");
    return frame_cnn_bwd(*d, *g, S(stream));
}
int mmqg_decoder_seq_fwd(const mmqg_decoder_seq* d, mmqg_stream stream) {
    MMQG_REQUIRE(d, "mmqg_decoder_seq_fwd: null descriptor");
    MMQG_TRY(persist_check_healthy("mmqg_decoder_seq_fwd"));
    return decoder_seq_fwd(*d, S(stream));
}
int mmqg_decoder_decode_run(const mmqg_decoder_decode* d, mmqg_stream stream) {
    MMQG_REQUIRE(d, "mmqg_decoder_decode_run: null descriptor");
    return decoder_decode(*d, S(stream));
}
int mmqg_sample_gumbel(const float* logits, int ld, int rows, int V, uint64_t seed, uint64_t stream_id, int64_t* out_ids,
                       mmqg_stream stream) {
    return sample_gumbel(logits, ld, rows, V, seed, stream_id, out_ids, S(stream));
}
int mmqg_decoder_seq_bwd(const mmqg_decoder_seq* d, const mmqg_decoder_seq_grad* g, mmqg_stream stream) {
    MMQG_REQUIRE(d && g, "mmqg_decoder_seq_bwd: null descriptor");
    MMQG_TRY(persist_check_healthy("mmqg_decoder_seq_bwd"));
    return decoder_seq_bwd(*d, *g, S(stream));
}

}  // extern "C"
