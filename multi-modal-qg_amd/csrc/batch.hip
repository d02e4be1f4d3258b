// One launch that turns a batch as the data layer hands it over (question-major, train.py:149-160) into
// the trainer's static input buffers: time-major frame features / raw frames, audio rows masked past
// n_frames inside the fused value tensor (train.py:156 pads with zeros), time-major context / target ids,
// the teacher-forcing decoder inputs (<start>, then target[t-1]: train.py:168,175), the per-row loss weights
// (t < target_len) / B, and the three length vectors, clamped to the static extents [0, Tc] / [0, Td] / [0, Tf]
// (the BatchNorm frame counts and the masks downstream trust them).  Replaces ~25 small framework kernels per step.
#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace {

__global__ __launch_bounds__(256) void pack_batch_kernel(mmqg_batch_pack a, int64_t n_frames_elems, int64_t n_audio, int64_t n_ctx,
                                                         int64_t n_tgt) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_frames_elems) {                       // feats[t][b][:] = frames[b][t][:]
        const int64_t j = i % a.frame_inner, tb = i / a.frame_inner;
        const int b = (int)(tb % a.B), t = (int)(tb / a.B);
        a.feats[i] = a.frames[((int64_t)b * a.Tf + t) * a.frame_inner + j];
        return;
    }
    i -= n_frames_elems;
    if (i < n_audio) {                              // value rows of the audio modality, zero past n_frames
        const int j = (int)(i % a.Da), t = (int)((i / a.Da) % a.audio_rows), b = (int)(i / ((int64_t)a.Da * a.audio_rows));
        const float v = t < min(a.n_frames[b], a.Tf) ? a.audio[i] : 0.f;
        a.audio_out[(int64_t)b * a.audio_stride_b + (int64_t)t * a.Da + j] = v;
        return;
    }
    i -= n_audio;
    if (i < n_ctx) {                                // ids_c[t][b] = context[b][t]
        const int b = (int)(i % a.B), t = (int)(i / a.B);
        a.ids_c[i] = a.context[(int64_t)b * a.Tc + t];
        return;
    }
    i -= n_ctx;
    if (i < n_tgt) {
        const int b = (int)(i % a.B), t = (int)(i / a.B);
        a.target_t[i] = a.target[(int64_t)b * a.Td + t];
        a.ids_d[i] = t == 0 ? a.start_id : a.target[(int64_t)b * a.Td + t - 1];
        a.row_w[i] = t < a.tgt_len[b] ? 1.0f / (float)a.B : 0.f;
        return;
    }
    i -= n_tgt;
    if (i < a.B) {
        a.ctx_len_out[i] = min(max(a.ctx_len[i], 0), a.Tc);
        a.tgt_len_out[i] = min(max(a.tgt_len[i], 0), a.Td);
        a.n_frames_out[i] = min(max(a.n_frames[i], 0), a.Tf);
    }
}

}  // namespace

namespace mmqg {

int pack_batch(const mmqg_batch_pack& a, hipStream_t s) {
    MMQG_REQUIRE(a.B >= 0 && a.Tf >= 0 && a.Tc >= 0 && a.Td >= 0 && a.Da >= 0 && a.audio_rows >= 0 && a.frame_inner >= 0,
                 "pack_batch: bad shape");
    if (a.B == 0) return 0;
    MMQG_REQUIRE(a.ctx_len && a.tgt_len && a.n_frames && a.ctx_len_out && a.tgt_len_out && a.n_frames_out, "pack_batch: null lengths");
    MMQG_REQUIRE(a.Tc == 0 || (a.context && a.ids_c), "pack_batch: null context");
    MMQG_REQUIRE(a.Td == 0 || (a.target && a.target_t && a.ids_d && a.row_w), "pack_batch: null target");
    const int64_t nf = a.frames ? (int64_t)a.Tf * a.B * a.frame_inner : 0;
    MMQG_REQUIRE(!a.frames || a.feats, "pack_batch: null feats");
    const int64_t na = a.audio ? (int64_t)a.B * a.audio_rows * a.Da : 0;
    MMQG_REQUIRE(!a.audio || a.audio_out, "pack_batch: null audio_out");
    const int64_t nc = (int64_t)a.Tc * a.B, nt = (int64_t)a.Td * a.B;
    const int64_t total = nf + na + nc + nt + a.B;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, s, a, nf, na, nc, nt);
    return check_launch("pack_batch");
}

}  // namespace mmqg
