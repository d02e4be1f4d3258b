// Host-callable launchers of the gfx950 kernels (internal C++ surface; the public boundary is
// include/mmqg.h).  Every function only enqueues work on the given stream: no allocation, no
// synchronisation, so a caller may capture any sequence of them into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmqg.h"

namespace mmqg {

// ---- gemm_f32.hip ---------------------------------------------------------------------
int gemm_f32(int a_layout, int b_layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
             const float* A2, int lda2, const float* B2, int ldb2, int K2, const float* bias, const float* bias2,
             int beta, float* C, int ldc, int split_k, hipStream_t s);
// gemm_nt_tile.hip: C = A[M,K] * B[N,K]^T + bias + bias2 for large outputs with short K (the vocabulary projection), one
// 256 x BN tile per CU; 0 = launched, 1 = not taken, < 0 = error
// stats (nullable, gemm_nt_stats_bytes(M, N) bytes): per row and column tile {max, sum exp(x - max), argmax index bits, 0}
// for ce_fwd_bwd; *stats_tiles = column tiles written per row (0 = none)
int64_t gemm_nt_stats_bytes(int M, int N);
int gemm_nt_tile(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias,
                 const float* bias2, float* C, int ldc, hipStream_t s, float* stats = nullptr, int64_t stats_bytes = 0,
                 int* stats_tiles = nullptr);
// independent accumulating products (C += A*B) of one layout in as few launches as possible
typedef mmqg_gemm_problem GemmProblem;
int gemm_f32_grouped(int a_layout, int b_layout, const GemmProblem* probs, int n, hipStream_t s);
// gemm_x3.hip: large products with fp32-exact operands split into bf16 pieces on the bf16 matrix cores.
// gemm_x3_grouped: 0 = launched, 1 = not taken (fall back), < 0 = error; bias arrays nullable (per problem, per column)
bool gemm_x3_wants(int M, int N, int K);
int gemm_x3_slots();
int gemm_x3_projection(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias, float* C,
                       int ldc, float* stats, int64_t stats_bytes, int* stats_tiles, hipStream_t s);
// colsum / colsum2 (nullable arrays, nullable entries; m-major A only): out[m] += sum over k of A[k][m]
int gemm_x3_grouped(int a_layout, int b_layout, const GemmProblem* probs, const float* const* bias, const float* const* bias2,
                    int n, hipStream_t s, float* const* colsum = nullptr, float* const* colsum2 = nullptr);
// weight-gradient group C_i += A_i^T B_i (both m/n-major) that also adds the column sums of A_i (the bias gradients) into
// colsum_i / colsum2_i where those are given: fused into the split-bf16 product's staging pass when that kernel takes
// the problem, a separate sweep otherwise
int gemm_f32_wgrad_group(const GemmProblem* probs, float* const* colsum, float* const* colsum2, int n, hipStream_t s);

// ---- lstm_cell.hip --------------------------------------------------------------------
struct CellFwd {
    int B, H;
    float* gates; int ld_g;                 // [B][4H] in: summed pre-activations (i,f,g,o blocks);
                                            // out: sigma(i),sigma(f),tanh(g),sigma(o); 0 for finished rows
    const float* h_prev; const float* c_prev;   // [B][H]
    float* h_out; float* c_out;             // [B][H]
    float* h_drop;                          // nullable [B][H]: h_out * dropout scale (input of the next layer)
    float* y_out; int64_t y_stride_b;       // nullable: y[b*stride + j] = row active ? h_out : 0
    const int32_t* lens; int t;             // nullable: row b is active while t < lens[b]
    float p; uint64_t seed; uint64_t stream_id;
    const int32_t* seed_off;                // nullable device word mixed into the seed at run time
};
int lstm_cell_fwd(const CellFwd& a, hipStream_t s);

struct CellBwd {
    int B, H;
    const float* gates_act;                 // [B][4H]
    const float* c_prev; const float* c_new;    // [B][H]
    float* dh_rec;                          // in: dL/dh from step t+1; out: pass-through part for step t-1
    const float* dh_above; int64_t above_stride_b;   // nullable: gradient of this layer's (dropped) output
    float p; uint64_t seed; uint64_t stream_id;       // dropout that was applied to that output
    const int32_t* seed_off;
    const float* dh_extra; int64_t extra_stride_b;   // nullable: further gradient of h (only for active rows)
    const float* dh_pre;                    // nullable [B][H]: part of the recurrent product formed ahead of time
                                            // (fused skinny kernel only; added for every row)
    float* dc;                              // in/out [B][H]
    float* dgates; int ld_dg;               // out [B][4H]
    const int32_t* lens; int t;
};
int lstm_cell_bwd(const CellBwd& a, hipStream_t s);

int dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t stream_id, const int32_t* seed_off,
                 hipStream_t s);

// ---- skinny.hip: batch-sized products fused with the step's pointwise epilogue ---------
struct SkinnyPair {
    const float* A; int lda;      // [M][K] k-major
    const float* B; int ldb;      // [N][K] k-major
    int K;
    int masked;                   // backward only: product goes through the layer's dropout mask
};
bool skinny_usable(const SkinnyPair* pairs, int npairs);
// workspace of the wide (64 x 32 tiles, k slices meeting at a ticket) backward / plain kernel for batches over 64 rows:
// set by an executor for the duration of its call, zero tickets at the end of the buffer
void skinny_set_wide_ws(float* ws, int64_t bytes);
int64_t skinny_wide_ws_bytes(int M, int max_N);
int skinny_plain(int M, int N, const SkinnyPair* pairs, int npairs, const float* bias, int beta, float* C, int ldc,
                 hipStream_t s);
// gates (+= if gates_has_pre) sum of pairs + bias1 + bias2, then the cell update of CellFwd
int skinny_cell_fwd(const SkinnyPair* pairs, int npairs, int gates_has_pre, const float* bias1, const float* bias2,
                    const CellFwd& f, hipStream_t s);
// dh(t) = sum of pairs + carry (f.dh_rec) [+ above, extra], then the cell backward of CellBwd
int skinny_cell_bwd(const SkinnyPair* pairs, int npairs, const CellBwd& f, hipStream_t s);
// up to three independent layer-steps (one wavefront diagonal of a layer stack) in ONE launch
struct SkinnyFwdJob { SkinnyPair pairs[3]; int npairs; int gates_has_pre; const float* bias1; const float* bias2; CellFwd cell; };
struct SkinnyBwdJob { SkinnyPair pairs[3]; int npairs; CellBwd cell; };
struct SkinnyPlainJob { int M, N; SkinnyPair pairs[3]; int npairs; const float* bias; int beta; float* C; int ldc; };
// up to 3 independent plain products in ONE launch
int skinny_plain_multi(const SkinnyPlainJob* jobs, int njobs, hipStream_t s);
// one cell-backward layer-step plus up to 2 independent plain products in ONE launch
int skinny_cell_bwd_plus(const SkinnyBwdJob& cell, const SkinnyPlainJob* extra, int nextra, hipStream_t s);
int skinny_cell_fwd_multi(const SkinnyFwdJob* jobs, int njobs, hipStream_t s);
// one cell-forward layer-step plus up to 2 independent plain products in ONE launch
int skinny_cell_fwd_plus(const SkinnyFwdJob& cell, const SkinnyPlainJob* extra, int nextra, hipStream_t s);
int skinny_cell_bwd_multi(const SkinnyBwdJob* jobs, int njobs, hipStream_t s);
int transpose_f32(const float* src, int ld_src, int rows, int cols, float* dst, int ld_dst, hipStream_t s);
int transpose_f32_batch(const mmqg_transpose_job* jobs, int n, hipStream_t s);

// ---- attention.hip --------------------------------------------------------------------
int attn_softmax_context_fwd(const mmqg_attn_values& v, const float* scores, int ld_s, float* attn, int ld_a,
                             float* ctx, int ld_c, hipStream_t s);
int attn_context_bwd(const mmqg_attn_values& v, const float* attn, int ld_a, const float* dctx, int ld_c,
                     const float* dattn, int ld_da, float* dscores, int ld_ds, hipStream_t s);
int attn_context_bwd_fused(const mmqg_attn_values& v, const float* attn, int ld_a, const float* ctx, int ld_x,
                           const float* dctx, int ld_c, float* dscores, int ld_ds, hipStream_t s);
// dV[rows < n_rows] = sum_t attn[t][b][seg+row] * dctx[t][b][off..off+D)
int attn_dvalues(int T, int B, int n_rows, int D, const float* attn, int64_t attn_stride_t, int ld_a, int seg_off,
                 const float* dctx, int64_t dctx_stride_t, int ld_c, int ctx_off, float* out, int64_t out_stride_row,
                 int64_t out_stride_b, int accumulate, hipStream_t s);

// attention_fused.hip: score product + softmax + context in one launch (0 = launched, 1 = not taken, < 0 = error)
int64_t attn_fused_ws_bytes(const mmqg_attn_values& v, int Hq);
int attn_fused_fwd(const mmqg_attn_values& v, const float* pre, int ld_s, const float* h, int ld_h, const float* W, int ld_w,
                   int Hq, float* attn, int ld_a, float* ctx, int ld_c, float* ws, int64_t ws_bytes, hipStream_t s);

// ---- embedding.hip --------------------------------------------------------------------
int embedding_fwd(const float* table, const int64_t* ids, float* out, int n, int V, int E, int ld_out, hipStream_t s);
int embedding_bwd(const float* dout, int ld, const int64_t* ids, float* dtable, int n, int V, int E, hipStream_t s);

// ---- loss.hip -------------------------------------------------------------------------
int ce_fwd_bwd(const float* logits, int ld, const int64_t* target, const float* row_weight, int rows, int V,
               float* loss_rows, int64_t* argmax, float* dlogits, int ld_d, hipStream_t s, const float* stats = nullptr,
               int stats_tiles = 0);
int colsum_add(const float* X, int ld, int M, int N, float* out, hipStream_t s);
// out1 += column sums, out2 += the same sums (the two bias gradients of an LSTM layer); out2 may be null
int colsum_add2(const float* X, int ld, int M, int N, float* out1, float* out2, hipStream_t s);
int sample_gumbel(const float* logits, int ld, int rows, int V, uint64_t seed, uint64_t stream_id, int64_t* out_ids,
                  hipStream_t s);
int fill_i64(int64_t* dst, int64_t value, int64_t n, hipStream_t s);
int reduce_sum(const float* x, int n, float* out, hipStream_t s);
int copy_or_zero_f32(float* dst, const float* src, int64_t n, hipStream_t s);
constexpr int kMaxCopySegs = 16;
struct CopySeg { float* dst; const float* src; int64_t n; };
struct CopyBatch { CopySeg seg[kMaxCopySegs]; };
// dst_i[0:n_i] = src_i ? src_i[0:n_i] : 0 for every segment, one launch per 16 segments
int copy_or_zero_multi(const CopySeg* segs, int n, hipStream_t s);
int axpy(float* y, const float* x, float alpha, int64_t n, hipStream_t s);
int add_rows_strided(float* dst, int64_t dst_stride, const float* src, int64_t src_stride, int rows, int cols,
                     hipStream_t s);

// ---- persist.hip: the forward time loop of an LSTM stack as one persistent launch ------
bool lstm_persist_shape_ok(int T, int B, int L, int H);
int64_t lstm_persist_ws_bytes(int T, int B, int L, int H);
// 0 = done, 1 = not eligible (caller falls back to one launch per diagonal), < 0 = error
int lstm_seq_fwd_persistent(const mmqg_lstm_seq& d, hipStream_t s);
int persist_launch_count();
// persist_bwd.hip: the backward time loop of an LSTM stack as one persistent launch
bool lstm_persist_bwd_shape_ok(int T, int B, int L, int H);
int64_t lstm_persist_bwd_ws_bytes(int T, int B, int L, int H);
int lstm_seq_bwd_persistent(const mmqg_lstm_seq& d, const mmqg_lstm_seq_grad& g, const mmqg_lstm_seq* d2,
                            const mmqg_lstm_seq_grad* g2, hipStream_t s);
int persist_bwd_launch_count();
void persist_bwd_set_trace(unsigned long long* buf, int64_t words);
// persist_dec.hip: the forward time loop of the attention decoder as one persistent launch
bool decoder_persist_shape_ok(const mmqg_decoder_seq& d);
int64_t decoder_persist_ws_bytes(const mmqg_decoder_seq& d);
int decoder_seq_fwd_persistent(const mmqg_decoder_seq& d, hipStream_t s);   // 0 done, 1 not taken, < 0 error
int decoder_persist_launch_count();
void decoder_persist_set_trace(unsigned long long* buf, int64_t words);
// persist_dec_bwd.hip: the backward time loop of the attention decoder as one persistent launch
int64_t decoder_persist_bwd_ws_bytes(const mmqg_decoder_seq& d, int ld_ds);
int decoder_seq_bwd_persistent(const mmqg_decoder_seq& d, const mmqg_decoder_seq_grad& g, hipStream_t s);   // 0 done, 1 not taken, < 0 error
int decoder_persist_bwd_launch_count();
void decoder_persist_bwd_set_trace(unsigned long long* buf, int64_t words);
// persist_rt.hip: who may launch a persistent kernel, and how a failed one reaches the host
void persist_runtime_prepare();
int persist_device_cus();
int persist_usable_cus(hipStream_t s, bool can_shrink);   // device CUs, or the stream's / process's CU mask; minus the reserve if can_shrink
void persist_set_reserved_cus(int n);
int persist_reserved_cus();
unsigned* persist_host_fail_word();
int persist_begin(hipStream_t s);        // 0 = granted, 1 = declined (another persistent launch may be in flight)
void persist_end(hipStream_t s);
int persist_declined_count();
int persist_failures();
void persist_clear_failures();
int persist_check_healthy(const char* who);
void persist_set_test_fault(int extra_workgroups, unsigned max_spins);
int persist_test_extra_wg();
unsigned persist_test_max_spins();

// ---- cnn.hip --------------------------------------------------------------------------
int frame_cnn_fwd(const mmqg_frame_cnn& d, hipStream_t s);
int frame_cnn_bwd(const mmqg_frame_cnn& d, const mmqg_frame_cnn_grad& g, hipStream_t s);

// ---- batch.hip ----
int pack_batch(const mmqg_batch_pack& a, hipStream_t s);
int fetch_mapped(const mmqg_copy_seg* segs, int n, hipStream_t s);

// ---- adam.hip -------------------------------------------------------------------------
int adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps,
              const int32_t* step, float grad_scale, hipStream_t s, const int32_t* skip = nullptr);
int persist_guard_refresh(int32_t* flag, hipStream_t s);
int counter_add(int32_t* ctr, int delta, hipStream_t s);

}  // namespace mmqg
