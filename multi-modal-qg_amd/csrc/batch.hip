// One launch that turns a batch as the data layer hands it over (question-major, train.py:149-160) into
// the trainer's static input buffers: time-major frame features / raw frames, audio rows masked past
// n_frames inside the fused value tensor (train.py:156 pads with zeros), time-major context / target ids,
// the teacher-forcing decoder inputs (<start>, then target[t-1]: train.py:168,175), the per-row loss weights
// (t < target_len) / B, and the three length vectors, clamped to the static extents [0, Tc] / [0, Td] / [0, Tf]
// (the BatchNorm frame counts and the masks downstream trust them).  Replaces ~25 small framework kernels per step.
#include <algorithm>

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

// A batch that starts in HOST memory (what the reference's DataLoader hands over, train.py:144-162): the kernel reads
// the pinned, device-mapped host buffers itself — 16 bytes per lane over PCIe — into device staging buffers.  No
// copy-engine call sits between two replays of the step graph (an asynchronous hipMemcpy there cost ~18 ms per step on
// this stack), the launch can run on a second stream beside the previous step, and it is one
// kernel for all seven tensors of a batch.  Few workgroups: the transfer is PCIe-bound, not CU-bound.
constexpr int kFetchSegs = 8;
struct FetchBatch { void* dst[kFetchSegs]; const void* src[kFetchSegs]; int64_t bytes[kFetchSegs]; int n; };

__global__ __launch_bounds__(256) void fetch_mapped_kernel(FetchBatch f) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
#pragma unroll
    for (int s = 0; s < kFetchSegs; ++s) {
        if (s >= f.n) break;
        const int64_t nb = f.bytes[s];
        const bool vec = ((reinterpret_cast<uintptr_t>(f.dst[s]) | reinterpret_cast<uintptr_t>(f.src[s])) & 15) == 0;
        if (vec) {
            const uint4* src = reinterpret_cast<const uint4*>(f.src[s]);
            uint4* dst = reinterpret_cast<uint4*>(f.dst[s]);
            const int64_t n16 = nb >> 4;
            for (int64_t i = tid; i < n16; i += 2 * stride) {           // two loads in flight per lane
                const int64_t j = i + stride;
                const uint4 a = src[i];
                uint4 b = a;
                if (j < n16) b = src[j];
                dst[i] = a;
                if (j < n16) dst[j] = b;
            }
            for (int64_t i = (n16 << 4) + tid; i < nb; i += stride)
                reinterpret_cast<unsigned char*>(f.dst[s])[i] = reinterpret_cast<const unsigned char*>(f.src[s])[i];
        } else {
            for (int64_t i = tid; i < nb; i += stride)
                reinterpret_cast<unsigned char*>(f.dst[s])[i] = reinterpret_cast<const unsigned char*>(f.src[s])[i];
        }
    }
}

}  // namespace

namespace mmqg {

int fetch_mapped(const mmqg_copy_seg* segs, int n, hipStream_t s) {
    MMQG_REQUIRE(n >= 0 && n <= kFetchSegs && (n == 0 || segs), "fetch_mapped: between 0 and %d segments", kFetchSegs);
    FetchBatch f{};
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        MMQG_REQUIRE(segs[i].bytes >= 0 && (segs[i].bytes == 0 || (segs[i].dst && segs[i].src)), "fetch_mapped: bad segment %d", i);
        f.dst[i] = segs[i].dst; f.src[i] = segs[i].src; f.bytes[i] = segs[i].bytes;
        total += segs[i].bytes;
    }
    f.n = n;
    if (total == 0) return 0;
    const int blocks = (int)std::min<int64_t>(64, std::max<int64_t>(1, total / (256 * 32)));
    hipLaunchKernelGGL(fetch_mapped_kernel, dim3(blocks), dim3(256), 0, s, f);
    return check_launch("fetch_mapped");
}

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
