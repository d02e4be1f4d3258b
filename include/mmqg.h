/* mmqg.h — C ABI of the MI355X (gfx950) hot path of ksg14/multi-modal-qg.
 *
 * The reference is pure Python on PyTorch CPU ops and has no FFI of its own; the arithmetic
 * this library replaces is the set of ATen calls made by model/decoder.py:74-107,
 * model/encoder.py:58-71,95-100 and the step driver train.py:149-181.  Each entry point below
 * names the reference lines it stands in for.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - the CALLER owns every buffer; pointers are raw device addresses (torch
 *     tensor.data_ptr()), fp32 row-major unless stated, ids int64, lengths int32;
 *   - the library never allocates, frees or synchronises; every function only enqueues
 *     kernels on the hipStream_t passed as `stream` (zero fills and copies are kernels too:
 *     hipMemsetAsync nodes proved racy under hipGraph replay), so autograd ordering on
 *     PyTorch's current stream holds and a call sequence can be captured into a hipGraph;
 *   - return value 0 = enqueued, negative = error; mmqg_last_error() returns the
 *     thread-local message (the Python side raises RuntimeError, as the reference's torch
 *     calls would);
 *   - gradient outputs named dw_* / db_* are ACCUMULATED into (+=), everything else is overwritten;
 *   - attention segments are always stacked text | audio | video (widths Lt, Lav, Lav), the
 *     order of the concat at decoder.py:99 and of the return tuple at decoder.py:107.
 */
#ifndef MMQG_H
#define MMQG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMQG_ABI_VERSION 10
#define MMQG_MAX_LAYERS 8

typedef void* mmqg_stream; /* hipStream_t */

enum { MMQG_K_MAJOR = 0, MMQG_MN_MAJOR = 1 };
enum { MMQG_MASK_REFERENCE_NOOP = 0, MMQG_MASK_INTENDED = 1 };

int mmqg_abi_version(void);
const char* mmqg_last_error(void);

/* ------------------------------------------------------------------------------------------
 * GEMM on the fp32 matrix cores.  C = (beta?C:0) + bias + bias2 + A*B [+ A2*B2].
 * Replaces torch.nn.Linear / the gate matmuls inside nn.LSTM (decoder.py:64-70,78,84,92,104,106;
 * encoder.py:54,91) and their autograd.  Layouts: MMQG_K_MAJOR = k contiguous (A[m*lda+k],
 * B[n*ldb+k] i.e. a torch [out,in] weight used as x*W^T); MMQG_MN_MAJOR = m / n contiguous
 * (A[k*lda+m], B[k*ldb+n]).  split_k: 1 = none, >1 = that many K slices combined with f32
 * atomics, <0 = let the library choose. */
int mmqg_gemm_f32(int a_layout, int b_layout, int M, int N, int K,
                  const float* A, int lda, const float* B, int ldb,
                  const float* A2, int lda2, const float* B2, int ldb2, int K2,
                  const float* bias, const float* bias2, int beta, float* C, int ldc,
                  int split_k, mmqg_stream stream);

/* Several independent products C_i = beta_i*C_i + A_i*B_i of ONE layout pair.  When every product
 * accumulates (beta = 1), is m-major x n-major (the weight gradients dW += dY^T X of a layer stack:
 * train.py:177 loss.backward()) and reaches the 128x128 tile, they run as one launch whose K slices
 * add atomically; otherwise they are issued one by one (same results either way). */
typedef struct {
    int32_t M, N, K;
    const float* A; int32_t lda;
    const float* B; int32_t ldb;
    float* C; int32_t ldc;
    int32_t beta;
} mmqg_gemm_problem;
int mmqg_gemm_f32_grouped(int a_layout, int b_layout, const mmqg_gemm_problem* problems, int n,
                          mmqg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Embedding lookup and its dense gradient (nn.Embedding shared by encoder.py:96 and
 * decoder.py:75; train.py:25-31).  ids outside [0,V) yield a zero row / are skipped. */
int mmqg_embedding_fwd(const float* table, const int64_t* ids, float* out, int n, int V, int E,
                       int ld_out, mmqg_stream stream);
int mmqg_embedding_bwd(const float* dout, int ld_dout, const int64_t* ids, float* dtable, int n,
                       int V, int E, mmqg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Decoder attention (decoder.py:78-95).  The value tensors of one question are
 * text [Lt][H], audio [Lav][Da], video [Lav][Dv]; *_stride_b is the element distance between
 * consecutive questions, so three separate tensors and one fused (text|audio|video) allocation
 * per question are both expressible.  text_len / av_len are read only when
 * mask_mode == MMQG_MASK_INTENDED (the reference's masks at decoder.py:79,85,93 are no-ops). */
typedef struct {
    int32_t B, Lt, Lav, H, Da, Dv;
    const float* text;  int64_t text_stride_b;
    const float* audio; int64_t audio_stride_b;
    const float* video; int64_t video_stride_b;
    const int32_t* text_len;
    const int32_t* av_len;
    int32_t mask_mode;
    int32_t zero_past_len;    /* 1: the caller guarantees that value rows at and past text_len[b] / av_len[b] are all
                                 zero (train.py:156-160 pads the encoder outputs with zeros) — the kernels then skip
                                 those rows instead of streaming them; results are identical (a zero row contributes
                                 nothing to a context and has a zero weight gradient).  0 (default): every row is read,
                                 as the reference's bmm does.  Needs text_len and av_len. */
} mmqg_attn_values;

/* scores [B][ld_s] (pre-softmax, Lt+2*Lav wide) -> attn [B][ld_a] (must NOT alias scores) and
 * ctx [B][ld_c] = text ctx (H) | audio ctx (Da) | video ctx (Dv). */
int mmqg_attn_softmax_context_fwd(const mmqg_attn_values* v, const float* scores, int ld_s,
                                  float* attn, int ld_a, float* ctx, int ld_c, mmqg_stream stream);
/* The training loop's form: score product + softmax + context in ONE launch (decoder.py:78-95 for a teacher-forced
 * step whose embedded-word half of the scores is hoisted).  scores = pre [B][ld_s] (hoisted half + bias, left
 * untouched) + h [B][ld_h] (Hq wide: h_top of the previous step) x W^T, W = the recurrent half of the stacked score
 * matrix, row s at W + s * ld_w (Hq floats); attn / ctx as above.  ws: mmqg_attn_fused_ws_bytes(v, Hq) bytes, 16-byte
 * aligned, ZERO-FILLED ONCE by the caller (tickets at its end; the kernel leaves them zero).  Returns 0 = launched,
 * 1 = shape / alignment not taken (run the score product and mmqg_attn_softmax_context_fwd instead), < 0 = error. */
int64_t mmqg_attn_fused_ws_bytes(const mmqg_attn_values* v, int Hq);
int mmqg_attn_scores_softmax_context_fwd(const mmqg_attn_values* v, const float* pre, int ld_s, const float* h, int ld_h,
                                         const float* W, int ld_w, int Hq, float* attn, int ld_a, float* ctx, int ld_c,
                                         float* ws, int64_t ws_bytes, mmqg_stream stream);
/* dscores[b] = softmax'(attn[b]) applied to d(attn) = values . dctx (+ dattn, the gradient of
 * the returned attention weights themselves, nullable) */
int mmqg_attn_context_bwd(const mmqg_attn_values* v, const float* attn, int ld_a,
                          const float* dctx, int ld_c, const float* dattn, int ld_da,
                          float* dscores, int ld_ds, mmqg_stream stream);
/* the same for callers that kept the forward's contexts ctx [B][ld_x]: the softmax Jacobian's row dot
 * sum_j attn_j d(attn)_j equals ctx . dctx per modality, so ONE kernel writes dscores (no dattn term) */
int mmqg_attn_context_bwd_fused(const mmqg_attn_values* v, const float* attn, int ld_a, const float* ctx, int ld_x,
                                const float* dctx, int ld_c, float* dscores, int ld_ds, mmqg_stream stream);
/* gradient of the first n_rows value rows of one modality, summed over T steps:
 * out[b][row][:] (+)= sum_t attn[t][b][seg_off+row] * dctx[t][b][ctx_off : ctx_off+D] */
int mmqg_attn_dvalues(int T, int B, int n_rows, int D,
                      const float* attn, int64_t attn_stride_t, int ld_a, int seg_off,
                      const float* dctx, int64_t dctx_stride_t, int ld_c, int ctx_off,
                      float* out, int64_t out_stride_row, int64_t out_stride_b, int accumulate,
                      mmqg_stream stream);

/* ------------------------------------------------------------------------------------------
 * LSTM cell (the pointwise part of nn.LSTM, gate order i,f,g,o).  gates [B][4H] holds the
 * summed pre-activations on entry and the activated gates on exit (kept for backward).
 * lens/t: row b is updated only while t < lens[b] (ragged batches), otherwise its state is
 * carried.  h_drop (nullable) receives h_out times the inter-layer dropout scale, a pure
 * function of (seed, stream_id, b*H+j) — see mmqg_dropout_mask. */
int mmqg_lstm_cell_fwd(int B, int H, float* gates, int ld_g, const float* h_prev, const float* c_prev,
                       float* h_out, float* c_out, float* h_drop, float* y_out, int64_t y_stride_b,
                       const int32_t* lens, int t, float dropout_p, uint64_t seed, uint64_t stream_id,
                       mmqg_stream stream);
int mmqg_lstm_cell_bwd(int B, int H, const float* gates_act, const float* c_prev, const float* c_new,
                       float* dh_rec, const float* dh_above, int64_t above_stride_b, float dropout_p,
                       uint64_t seed, uint64_t stream_id, const float* dh_extra, int64_t extra_stride_b,
                       float* dc, float* dgates, int ld_dg, const int32_t* lens, int t,
                       mmqg_stream stream);
/* out[i] = 0 or 1/(1-p): the scale the kernels apply to element i of stream `stream_id`.
 * seed_offset (nullable, also a field of the sequence descriptors): a device int32 that is
 * mixed into the seed when the kernel RUNS, so a captured graph that is replayed with the
 * step counter advanced draws fresh masks. */
int mmqg_dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t stream_id,
                      const int32_t* seed_offset, mmqg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Cross entropy over vocabulary logits (train.py:174,264) with the argmax used by the greedy
 * decode (train.py:107-108).  loss_rows[r] = row_weight[r] * CE(logits[r], target[r]);
 * dlogits (nullable, may alias logits) = row_weight[r] * (softmax - onehot). */
int mmqg_ce_fwd_bwd(const float* logits, int ld, const int64_t* target, const float* row_weight,
                    int rows, int V, float* loss_rows, int64_t* argmax, float* dlogits, int ld_d,
                    mmqg_stream stream);
/* The vocabulary projection of the teacher-forced step (decoder.py:106 nn.Linear(hidden, vocab) over all
 * Td*B rows at once) with the loss's row statistics taken in the product's epilogue: when the one-tile-per-CU
 * kernel serves the shape, stats[r][t] = {max, sum exp(x - max), first argmax (int bits), 0} over column tile t
 * of row r and *stats_tiles (host int, written before the call returns) = tiles per row; otherwise the generic
 * product runs and *stats_tiles = 0.  mmqg_ce_fwd_bwd_stats then combines the tiles instead of sweeping the
 * logits for max and sum-exp (stats_tiles == 0: identical to mmqg_ce_fwd_bwd).  stats (nullable) holds
 * mmqg_projection_stats_ws_bytes(rows, V) bytes, 16-byte aligned. */
int64_t mmqg_projection_stats_ws_bytes(int rows, int V);
/* which kernel the last mmqg_projection_fwd call took: 0 generic tiled fp32 MFMA, 1 one-tile-per-CU fp32 MFMA,
 * 2 split-bf16 (fp32-exact operands as three bf16 pieces each, six bf16 MFMAs per product, fp32 accumulation) */
int mmqg_projection_last_kernel(void);
int mmqg_projection_fwd(int rows, int V, int H, const float* h, int ldh, const float* W, int ldw,
                        const float* bias, float* logits, int ld, float* stats, int64_t stats_bytes,
                        int32_t* stats_tiles, mmqg_stream stream);
int mmqg_ce_fwd_bwd_stats(const float* logits, int ld, const int64_t* target, const float* row_weight,
                          int rows, int V, const float* stats, int stats_tiles, float* loss_rows,
                          int64_t* argmax, float* dlogits, int ld_d, mmqg_stream stream);
/* Weight and bias gradient of a Linear layer y = x W^T + b (autograd of decoder.py:106, the vocabulary projection):
 * dW[out][in] += dY^T X over `rows` rows, dbias[out] += column sums of dY (nullable).  dY is [rows][out] (ld_dy),
 * X is [rows][in] (ldx).  The column sums ride in the product's operand staging pass when the split-bf16 kernel
 * takes the shape; otherwise they are a sweep of their own. */
int mmqg_linear_wgrad(int out_features, int in_features, int rows, const float* dY, int ld_dy,
                      const float* X, int ldx, float* dW, int lddw, float* dbias, mmqg_stream stream);
int mmqg_colsum_add(const float* X, int ld, int M, int N, float* out, mmqg_stream stream);
int mmqg_reduce_sum(const float* x, int n, float* out, mmqg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam defaults, train.py:265-267,179-181) over a flat parameter range.
 * `step` is a device int32 holding the 1-based step number (so a captured graph can be
 * replayed); grads are multiplied by grad_scale first (1/world_size after an all-reduce sum). */
int mmqg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1,
                   double b2, double eps, const int32_t* step, float grad_scale, mmqg_stream stream);
/* The same with a guard: skip (nullable) is a device int32; when skip[0] != 0 the launch leaves p, m and v untouched.
 * mmqg_persist_guard_refresh(flag, stream) sets flag[0] = (a persistent time loop of this process has reported a failure,
 * mmqg_persist_failures() > 0) with ONE device thread, so that the optimizer launches behind it — also inside a replayed
 * graph, where the host cannot intervene — drop the update of a step whose gradients were poisoned with NaN. */
int mmqg_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2,
                           double eps, const int32_t* step, float grad_scale, const int32_t* skip, mmqg_stream stream);
int mmqg_persist_guard_refresh(int32_t* flag, mmqg_stream stream);
int mmqg_counter_add(int32_t* counter, int delta, mmqg_stream stream);
/* dst[c][r] = src[r][c]: used to keep k-major (transposed) copies of the recurrent weights for
 * the backward time loops; refresh after every optimizer step. */
int mmqg_transpose_f32(const float* src, int ld_src, int rows, int cols, float* dst, int ld_dst,
                       mmqg_stream stream);
/* several transposes in ONE launch (all recurrent weights of a model after an optimizer step) */
typedef struct { const float* src; int32_t ld_src; int32_t rows; int32_t cols; float* dst; int32_t ld_dst; } mmqg_transpose_job;
int mmqg_transpose_f32_batch(const mmqg_transpose_job* jobs, int n, mmqg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Whole-sequence executors: one call enqueues every kernel of a time loop.
 *
 * mmqg_lstm_seq: stacked LSTM over T steps, time-major, zero or given initial state.  Serves
 * TextEncoder driven by train.py:159-166 (L=3, input = embedded context tokens) and the LSTM
 * stage of VideoConvLstmEncoder (encoder.py:54,69; L=1).  Layer by layer: the input product
 * X*W_ih^T of a whole layer is one GEMM, only h*W_hh^T + cell stay in the time loop. */
typedef struct {
    int32_t T, B, L, H, In;
    const float* x; int32_t ldx;                    /* [T][B][In] */
    const float* w_ih[MMQG_MAX_LAYERS];             /* [4H][In or H] */
    const float* w_hh[MMQG_MAX_LAYERS];             /* [4H][H] */
    const float* b_ih[MMQG_MAX_LAYERS];
    const float* b_hh[MMQG_MAX_LAYERS];
    const float* w_hhT[MMQG_MAX_LAYERS];            /* optional [H][4H] transposed copies: enable the
                                                       fused one-launch-per-layer-step backward */
    const float* w_ihT[MMQG_MAX_LAYERS];            /* optional [H][4H] for l >= 1: with w_hhT, a multi-layer
                                                       backward runs as a wavefront over (layer, time) diagonals */
    const float* h0; const float* c0;               /* [L][B][H] or NULL = zeros */
    const int32_t* lens;                            /* [B] or NULL */
    float dropout_p; int32_t training; uint64_t seed; uint64_t stream_base;
    const int32_t* seed_offset;                     /* nullable, see mmqg_dropout_mask */
    float* gates;                                   /* [L][T][B][4H] */
    float* hs; float* cs;                           /* [L][T+1][B][H]; slot 0 = initial state */
    float* hdrop;                                   /* [L-1][T][B][H], NULL if no dropout */
    float* y; int64_t y_stride_t; int64_t y_stride_b; /* optional top-layer outputs (0 past lens) */
    float* persist_ws; int64_t persist_ws_bytes;    /* optional workspace (16-byte aligned, size from
                                                       mmqg_lstm_seq_persist_ws_bytes, ZERO-FILLED ONCE by the caller):
                                                       with it, and a shape the persistent kernels take, a whole time
                                                       loop (forward; backward too) is ONE launch whose workgroups keep
                                                       the recurrent weights in LDS and meet at device-wide barriers.
                                                       Two such launches must not be in flight on one device at once
                                                       (each could end up half resident and wait for workgroups the
                                                       other keeps off the CUs): the library grants a request only on
                                                       the stream of the previous persistent launch or once that one
                                                       has completed, and to one stream per capture; a declined request
                                                       takes the launch-per-diagonal path (same results).  What the
                                                       host cannot see (another process on the device, two graphs
                                                       replayed side by side) ends at a bounded spin: see
                                                       mmqg_persist_failures() */
} mmqg_lstm_seq;

typedef struct {
    const float* dy; int64_t dy_stride_t; int64_t dy_stride_b; /* grad of y, nullable */
    const float* dhT; const float* dcT;             /* [L][B][H] grad of the final state, nullable */
    float* dgates;                                  /* [L][T][B][4H] scratch */
    float* dxl;                                     /* [T][B][H] scratch */
    float* dh; float* dc;                           /* [L][B][H] scratch */
    float* dx; int32_t lddx;                        /* [T][B][In] out, nullable */
    float* dw_ih[MMQG_MAX_LAYERS]; float* dw_hh[MMQG_MAX_LAYERS];
    float* db_ih[MMQG_MAX_LAYERS]; float* db_hh[MMQG_MAX_LAYERS];
    float* dh0; float* dc0;                         /* [L][B][H] out, nullable */
    int32_t phase;                                  /* 0 = everything; 1 = the time loops only (+ the
                                                       inter-layer input gradients they need); 2 = weight and
                                                       bias gradients + dx only (after a phase-1 call).  Lets a
                                                       caller run the large, recurrence-free GEMMs of phase 2 on
                                                       a second stream beside another sequence's time loop. */
    float* persist_ws; int64_t persist_ws_bytes;    /* optional workspace of the persistent BACKWARD time loop (size
                                                       from mmqg_lstm_seq_bwd_persist_ws_bytes, 16-byte aligned,
                                                       zero-filled once by the caller): with it, and a shape that
                                                       kernel takes, all T + L - 1 anti-diagonals of the backward
                                                       wavefront are ONE launch (the recurrent weights cut along K
                                                       and kept in LDS, partial products meeting through an exchange
                                                       buffer behind a device-wide barrier).  Same residency rule and
                                                       failure reporting as mmqg_lstm_seq.persist_ws */
    float* wide_ws; int64_t wide_ws_bytes;          /* optional workspace for batches over 64 rows (size from
                                                       mmqg_wide_ws_bytes, 16-byte aligned, zero-filled once): the
                                                       backward layer-steps then run on 64 x 32 tiles whose k slices
                                                       meet at a ticket instead of 16 x 16 tiles */
} mmqg_lstm_seq_grad;

int mmqg_lstm_seq_fwd(const mmqg_lstm_seq* d, mmqg_stream stream);
/* bytes of persist_ws the persistent forward needs for this shape; 0 = shape not taken (B > 64, H < 128 or not a
 * multiple of 16, ...): the forward then runs one launch per wavefront diagonal */
int64_t mmqg_lstm_seq_persist_ws_bytes(int T, int B, int L, int H);
/* diagnostics: how many time loops this process has run as persistent launches so far (forward + backward), and how
 * many requests the in-flight guard has declined (those ran as one launch per diagonal) */
int mmqg_persist_launch_count(void);
int mmqg_persist_declined_count(void);
/* Health of the persistent time loops.  A launch whose device-wide barrier timed out (its workgroups were not all
 * resident) poisons its output with NaN, bumps a counter in its workspace and sets a word in pinned host memory that
 * this call reads WITHOUT synchronising (the one host allocation the library makes: 64 bytes, at the first
 * *_persist_ws_bytes query).  > 0: at least one launch failed; mmqg_lstm_seq_fwd/bwd and mmqg_decoder_seq_fwd/bwd then
 * return an error (mmqg_last_error() explains) until mmqg_persist_clear_failures().  A caller that replays captured
 * graphs makes no library call per step: it polls this function (BatchedTrainer.step does). */
int mmqg_persist_failures(void);
int mmqg_persist_clear_failures(void);
/* test hook: the following persistent launches wait for extra_workgroups more arrivals than their grid has and give up
 * after max_spins polls (0, 0 switches it off) — exercises the failure path without occupying the chip.  Refused
 * (returns -1) unless the process was started with MMQG_ENABLE_TEST_HOOKS=1: a product process cannot arm it. */
int mmqg_persist_set_test_fault(int extra_workgroups, uint32_t max_spins);
/* Data-parallel runs: a persistent launch needs ALL its workgroups resident, and an RCCL channel kernel that waits for
 * a slow peer holds its CUs meanwhile.  The persistent launches that can overlap a collective (the text encoder's
 * backward time loop: the first gradient buckets travel beside it) therefore size their grid to the usable CUs minus
 * n; the caller caps RCCL's channels to match (NCCL_MAX_NCHANNELS).  Launches that never overlap a collective (every
 * forward loop, the decoder's backward loop: no bucket is final before they end) keep the whole chip.  0 = off. */
int mmqg_persist_set_reserved_cus(int n);
/* CUs a persistent launch on `stream` would count on: the device's, or fewer under a CU mask of the stream
 * (hipExtStreamCreateWithCUMask) or of the process (ROC_GLOBAL_CU_MASK); shrinkable != 0: minus the reserve above */
int mmqg_persist_usable_cus(mmqg_stream stream, int shrinkable);
/* diagnostics: the following persistent launches write 4 wall-clock stamps (100 MHz) per (workgroup, diagonal) into
 * buf[words] (start, products done, cell done, barrier passed); NULL switches it off.  Stamped launches run a
 * separate instantiation of the kernel: the product kernel carries no stamps. */
int mmqg_persist_set_trace(uint64_t* buf, int64_t words);
int mmqg_lstm_seq_bwd(const mmqg_lstm_seq* d, const mmqg_lstm_seq_grad* g, mmqg_stream stream);
/* The backward TIME LOOPS (phase 1 of mmqg_lstm_seq_bwd) of two independent stacks in one call: (d, g) as above and a
 * second, single-layer stack (d2, g2) of the same B and H — the frame LSTM of VideoConvLstmEncoder (encoder.py:54,69)
 * beside the text encoder's (encoder.py:91), whose gradients both become available when the decoder's backward loop
 * ends (train.py:177).  When g->persist_ws is given and the persistent kernel takes the pair, both loops are ONE
 * launch (the second stack's T2 steps ride on the first T2 anti-diagonals); otherwise each loop runs on its own.
 * Phase 2 (weight gradients, dx) of each stack is the caller's next call.  g->phase / g2->phase are ignored. */
int mmqg_lstm_seq_bwd_pair(const mmqg_lstm_seq* d, const mmqg_lstm_seq_grad* g, const mmqg_lstm_seq* d2,
                           const mmqg_lstm_seq_grad* g2, mmqg_stream stream);
/* bytes of the wide_ws workspaces for a batch of B rows and products up to max_N columns wide; 0 = not used (B <= 64) */
int64_t mmqg_wide_ws_bytes(int B, int max_N);
/* bytes of mmqg_lstm_seq_grad.persist_ws for this shape; 0 = shape not taken (B > 64, H not a multiple of 64 or
 * below 128, weights beyond the chip's LDS, ...): the backward then runs one launch per anti-diagonal */
int64_t mmqg_lstm_seq_bwd_persist_ws_bytes(int T, int B, int L, int H);
/* diagnostics, as mmqg_persist_launch_count / mmqg_persist_set_trace for the backward kernel (6 stamps per
 * (workgroup, diagonal): start, partial products stored, first barrier passed, gate gradients published, arrived at
 * the second barrier, second barrier passed) */
int mmqg_persist_bwd_launch_count(void);
int mmqg_persist_bwd_set_trace(uint64_t* buf, int64_t words);

/* mmqg_decoder_seq: AttnDecoder.forward (decoder.py:74-107) for T teacher-forced steps
 * (train.py:171-175): per step attention scores, three softmaxes, three contexts, the L-layer
 * LSTM step.  Products that do not depend on the recurrence (the embedded-word part of the
 * scores and of the layer-0 gates) are hoisted into two GEMMs over all T*B rows; the vocabulary
 * projection (decoder.py:106) runs afterwards as one GEMM over hs[L-1][1..T]. */
typedef struct {
    int32_t T, B, L, H, E;
    mmqg_attn_values values;
    const float* xemb;                              /* [T][B][E] embedded input words */
    const float* w_attn; const float* b_attn;       /* [Lt+2Lav][E+H], [Lt+2Lav] */
    const float* w_ih[MMQG_MAX_LAYERS]; const float* w_hh[MMQG_MAX_LAYERS];
    const float* b_ih[MMQG_MAX_LAYERS]; const float* b_hh[MMQG_MAX_LAYERS];
    /* optional transposed copies (all or none): w_hhT[l] [H][4H]; w_ihT[l] [H][4H] for l >= 1;
     * w_ih0cT [H+Da+Dv][4H] = (W_ih0[:, E:])^T; w_attn_hT [H][ld_attn] = (W_attn[:, E:])^T, columns
     * past Lt+2Lav zero.  With them the backward time loop is one fused launch per layer-step. */
    const float* w_hhT[MMQG_MAX_LAYERS]; const float* w_ihT[MMQG_MAX_LAYERS];
    const float* w_ih0cT; const float* w_attn_hT;
    const float* h0; const float* c0;               /* [L][B][H] */
    const int32_t* lens;                            /* [B] target lengths or NULL */
    float dropout_p; int32_t training; uint64_t seed; uint64_t stream_base;
    const int32_t* seed_offset;
    float* scores;                                  /* [T][B][ld_attn] scratch: pre-softmax scores */
    float* attn; int32_t ld_attn;                   /* [T][B][ld_attn] attention weights (kept for backward) */
    float* ctx;                                     /* [T][B][H+Da+Dv] */
    float* gates;                                   /* [L][T][B][4H] */
    float* hs; float* cs;                           /* [L][T+1][B][H] */
    float* hdrop;                                   /* [L-1][T][B][H] or NULL */
    int32_t phase;                                  /* 0 = everything; 1 = hoisted products only (need only
                                                       xemb); 2 = state init + time loop (after phase 1) */
    int64_t h0_stride_l;                            /* elements between the layers of h0 / c0; 0 = B*H.  Lets h0/c0
                                                       point at the final slot of an encoder's hs/cs (train.py:169) */
    float* attn_ws; int64_t attn_ws_bytes;          /* optional workspace (mmqg_attn_fused_ws_bytes(&values, H), zero-filled
                                                       once): the per-step score product, softmax and contexts then run
                                                       as ONE launch (mmqg_attn_scores_softmax_context_fwd) */
    float* persist_ws; int64_t persist_ws_bytes;    /* optional workspace (mmqg_decoder_seq_persist_ws_bytes): the whole
                                                       time loop then runs as ONE persistent launch when the shape is
                                                       taken (MMQG_NO_PERSIST_DEC=1 / MMQG_NO_PERSIST=1: never) */
} mmqg_decoder_seq;

typedef struct {
    const float* dhtop;                             /* [T][B][H] grad of hs[L-1][1..T] */
    float* dgates;                                  /* [L][T][B][4H] */
    float* dscores; int32_t ld_ds;                  /* [T][B][ld_ds] */
    float* dctx;                                    /* [T][B][H+Da+Dv] */
    float* dh; float* dc;                           /* [L][B][H] scratch; on exit grad of h0/c0 */
    float* dxa;                                     /* [L][B][H] scratch */
    float* dxemb;                                   /* [T][B][E] out */
    float* dw_attn; float* db_attn;
    float* dw_ih[MMQG_MAX_LAYERS]; float* dw_hh[MMQG_MAX_LAYERS];
    float* db_ih[MMQG_MAX_LAYERS]; float* db_hh[MMQG_MAX_LAYERS];
    /* gradient of the leading value rows (the only ones an encoder produced) */
    int32_t n_text_rows;  float* dtext;  int64_t dtext_stride_row;  int64_t dtext_stride_b;
    int32_t n_video_rows; float* dvideo; int64_t dvideo_stride_row; int64_t dvideo_stride_b;
    int32_t phase;                                  /* 0 = everything; 1 = time loop + initial-state and value
                                                       gradients (what the encoders' backward needs); 2 = dxemb,
                                                       weight and bias gradients */
    /* optional [L][B][H] workspace: with it the time loop forms the recurrent product dgates_l(t) * W_hh_l
     * one step ahead of its use, as an extra job of a launch that is on the dependent chain anyway, so the
     * cell kernels of the chain carry only the operand pair that really is late. */
    float* dh_pre;
    float* wide_ws; int64_t wide_ws_bytes;          /* as mmqg_lstm_seq_grad.wide_ws (max_N = the widest product of the
                                                       loop: max(H, H + Da + Dv, ld_attn)) */
    float* persist_ws; int64_t persist_ws_bytes;    /* optional workspace (mmqg_decoder_seq_bwd_persist_ws_bytes, 16-byte
                                                       aligned; needs no initial value): the whole backward time loop
                                                       (phase 1 without the value gradients) then runs as ONE persistent
                                                       launch when the shape is taken and the transposed weight copies
                                                       are given (MMQG_NO_PERSIST_DEC_BWD=1 / MMQG_NO_PERSIST=1: never);
                                                       failure reporting as mmqg_lstm_seq.persist_ws */
} mmqg_decoder_seq_grad;

/* mmqg_decoder_decode: the free-running decode loop of validate() / evaluate()
 * (train.py:100-110, evaluate.py:70-103): every step feeds the token picked at the previous step
 * back through the embedding — all on the device, no host round trip per token.
 * strategy 0 = greedy argmax (train.py:107-108; evaluate.py 'greedy' and 'topk' with k=1),
 * strategy 1 = sampling from softmax(logits) (evaluate.py 'sampling'; Gumbel-max with the Philox
 * stream, so the draw differs from numpy's but has the same distribution).
 * Stopping at <end> (evaluate.py:101-103) is left to the caller: all T steps are produced. */
typedef struct {
    int32_t T, B, L, H, E, V;
    mmqg_attn_values values;
    const float* emb_table;                         /* [V][E] */
    const float* w_attn; const float* b_attn;       /* [Lt+2Lav][E+H] stacked text|audio|video */
    const float* w_ih[MMQG_MAX_LAYERS]; const float* w_hh[MMQG_MAX_LAYERS];
    const float* b_ih[MMQG_MAX_LAYERS]; const float* b_hh[MMQG_MAX_LAYERS];
    const float* w_out; const float* b_out;         /* [V][H], [V] */
    const float* h0; const float* c0;               /* [L][B][H] */
    int64_t start_id;
    int32_t strategy; uint64_t seed;
    const int64_t* target; const float* row_weight; /* optional [T][B]: per-step loss like validate() */
    int64_t* ids;                                   /* [T+1][B]: row 0 = start tokens, row t+1 = pick of step t */
    float* loss_rows;                               /* [T][B] or NULL */
    float* attn; int32_t ld_attn;                   /* [T][B][ld_attn] attention weights (returned) */
    float* xemb; float* scores; float* ctx;         /* scratch [B][E], [B][ld_attn], [B][H+Da+Dv] */
    float* gates;                                   /* scratch [L][B][4H] */
    float* hs; float* cs;                           /* scratch [2][L][B][H]; final state in slot T%2 */
    float* logits;                                  /* [T][B][V] if keep_logits else scratch [B][V] */
    int32_t keep_logits;
} mmqg_decoder_decode;

int mmqg_decoder_decode_run(const mmqg_decoder_decode* d, mmqg_stream stream);
int mmqg_sample_gumbel(const float* logits, int ld, int rows, int V, uint64_t seed, uint64_t stream_id,
                       int64_t* out_ids, mmqg_stream stream);

int mmqg_decoder_seq_fwd(const mmqg_decoder_seq* d, mmqg_stream stream);
/* bytes of mmqg_decoder_seq.persist_ws for this descriptor (T, B, L, H, E, values, ld_attn are read); 0 = the shape is
 * not taken (B > 64, L != 3, weights beyond the chip's LDS, ...) or MMQG_NO_PERSIST_DEC=1: the time loop then runs five
 * launches per token */
int64_t mmqg_decoder_seq_persist_ws_bytes(const mmqg_decoder_seq* d);
/* diagnostics, as mmqg_persist_launch_count / mmqg_persist_set_trace (8 stamps per (workgroup, token): start, scores
 * stored, barrier passed, contexts stored, barrier passed, layer 0 / 1 / 2 arrived) */
int mmqg_decoder_persist_launch_count(void);
int mmqg_decoder_persist_set_trace(uint64_t* buf, int64_t words);
int mmqg_decoder_seq_bwd(const mmqg_decoder_seq* d, const mmqg_decoder_seq_grad* g, mmqg_stream stream);
/* bytes of mmqg_decoder_seq_grad.persist_ws for this pair of descriptors (T, B, L, H, values, ld_attn, ld_ds are read);
 * 0 = the shape is not taken (B > 64, L != 3, H not a multiple of 32 or beyond 512, the three layers' recurrent weights
 * beyond the chip's LDS, ...) or MMQG_NO_PERSIST_DEC_BWD=1: the backward time loop then runs five launches per token */
int64_t mmqg_decoder_seq_bwd_persist_ws_bytes(const mmqg_decoder_seq* d, const mmqg_decoder_seq_grad* g);
/* diagnostics, as mmqg_decoder_persist_launch_count / _set_trace for the backward kernel (16 stamps per (workgroup, token)) */
int mmqg_decoder_persist_bwd_launch_count(void);
int mmqg_decoder_persist_bwd_set_trace(uint64_t* buf, int64_t words);

/* ------------------------------------------------------------------------------------------
 * Frame CNN of VideoConvLstmEncoder (encoder.py:40-50,64-67): blocks of 3x3 valid convolution
 * (stride 1) -> ReLU -> BatchNorm2d whose batch is the T frames of ONE question, optional 3x3/3
 * max-pool.  frames [B][T][Cin][H][W] ([T][B]... with time_major) is the layout after the reference's
 * view(T,C,H,W);
 * frames with t >= n_frames[b] are left out of the statistics and give zero features.
 * Training mode uses per-question batch statistics and advances running_mean/var once per
 * question in batch order; eval mode normalises with the running statistics. */
#define MMQG_CNN_MAX_BLOCKS 4
typedef struct {
    int32_t cout; int32_t pool;
    const float* w; const float* bias;              /* [cout][cin][3][3], [cout] */
    const float* gamma; const float* beta;          /* BatchNorm affine [cout] */
    float* running_mean; float* running_var;        /* [cout] */
    float* y;                                       /* [B*T][cout][h-2][w-2] relu(conv), kept for backward */
    float* z;                                       /* block output [B*T][cout][hz][wz] */
    uint8_t* argmax;                                /* pooled blocks: [B*T][cout][hz][wz] */
    double* stats;                                  /* [B][cout][2] scratch */
    float* mean; float* invstd; float* scale; float* shift;   /* [B][cout] */
} mmqg_cnn_block;

typedef struct {
    int32_t B, T, Cin, H, W, n_blocks, training;
    int32_t time_major;                             /* 0: frames and every block buffer are [B][T]...; 1: [T][B]... */
    float eps, momentum;
    const float* frames;
    const int32_t* n_frames;                        /* [B] or NULL */
    mmqg_cnn_block block[MMQG_CNN_MAX_BLOCKS];
} mmqg_frame_cnn;

typedef struct {
    const float* dfeat;                             /* gradient of the last block's output */
    float* dconv; float* dz;                        /* scratch: largest y / largest block input */
    float* dw[MMQG_CNN_MAX_BLOCKS]; float* dbias[MMQG_CNN_MAX_BLOCKS];     /* accumulated (+=) */
    float* dgamma[MMQG_CNN_MAX_BLOCKS]; float* dbeta[MMQG_CNN_MAX_BLOCKS];
} mmqg_frame_cnn_grad;

int mmqg_frame_cnn_fwd(const mmqg_frame_cnn* d, mmqg_stream stream);
int mmqg_frame_cnn_bwd(const mmqg_frame_cnn* d, const mmqg_frame_cnn_grad* g, mmqg_stream stream);

/* ------------------------------------------------------------------------------------------
 * One launch that repacks a question-major batch (what train.py:149-160 builds per question, here for
 * B questions) into the time-major static inputs of the sequence executors: frames / features
 * [B][Tf][inner] -> feats [Tf][B][inner] (frames NULL = already in place), audio [B][audio_rows][Da] ->
 * the audio rows of the fused value tensor with rows t >= n_frames[b] zeroed (train.py:156), context /
 * target ids [B][T] -> [T][B], decoder inputs ids_d[t] = t ? target[t-1] : start_id (train.py:168,175),
 * row_w[t][b] = (t < tgt_len[b]) / B, and copies of the three length vectors. */
typedef struct {
    int32_t B, Tf, Tc, Td, Da, audio_rows;
    int64_t frame_inner;
    const float* frames; const float* audio;
    const int64_t* context; const int64_t* target;
    const int32_t* ctx_len; const int32_t* tgt_len; const int32_t* n_frames;
    int64_t start_id;
    float* feats;
    float* audio_out; int64_t audio_stride_b;
    int64_t* ids_c; int64_t* ids_d; int64_t* target_t; float* row_w;
    int32_t* ctx_len_out; int32_t* tgt_len_out; int32_t* n_frames_out;
} mmqg_batch_pack;
int mmqg_pack_batch(const mmqg_batch_pack* a, mmqg_stream stream);
/* Host batches (train.py:144-162: the DataLoader hands every batch over in host memory).  src: PINNED host memory that is
 * mapped into the device's address space (hipHostMalloc, torch pin_memory()); the kernel reads it itself over PCIe with
 * 16-byte loads and writes dst (device).  At most 8 segments per call, ONE launch, no copy-engine call: it can sit on a
 * second stream beside the previous step's graph replay.  The caller keeps src unchanged until the launch has run. */
typedef struct { void* dst; const void* src; int64_t bytes; } mmqg_copy_seg;
int mmqg_fetch_mapped(const mmqg_copy_seg* segs, int n, mmqg_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MMQG_H */
