"""ctypes binding of ``libmmqg_hip.so`` (the C ABI declared in ``include/mmqg.h``).

The library is the product path: if it cannot be loaded, or a call is attempted with
tensors that are not on a ROCm device, this module raises — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmmqg_hip.so")
ABI_VERSION = 10
MAX_LAYERS = 8

K_MAJOR, MN_MAJOR = 0, 1
MASK_REFERENCE_NOOP, MASK_INTENDED = 0, 1

c_f = C.c_void_p          # device pointers travel as void* (tensor.data_ptr())
c_i = C.c_int
c_i64 = C.c_int64
c_u64 = C.c_uint64
c_fl = C.c_float


class AttnValues(C.Structure):
    _fields_ = [("B", C.c_int32), ("Lt", C.c_int32), ("Lav", C.c_int32), ("H", C.c_int32), ("Da", C.c_int32),
                ("Dv", C.c_int32),
                ("text", c_f), ("text_stride_b", c_i64),
                ("audio", c_f), ("audio_stride_b", c_i64),
                ("video", c_f), ("video_stride_b", c_i64),
                ("text_len", c_f), ("av_len", c_f), ("mask_mode", C.c_int32),
                ("zero_past_len", C.c_int32)]


_PTRS = c_f * MAX_LAYERS


class LstmSeq(C.Structure):
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("In", C.c_int32),
                ("x", c_f), ("ldx", C.c_int32),
                ("w_ih", _PTRS), ("w_hh", _PTRS), ("b_ih", _PTRS), ("b_hh", _PTRS), ("w_hhT", _PTRS),
                ("w_ihT", _PTRS),
                ("h0", c_f), ("c0", c_f), ("lens", c_f),
                ("dropout_p", c_fl), ("training", C.c_int32), ("seed", c_u64), ("stream_base", c_u64),
                ("seed_offset", c_f),
                ("gates", c_f), ("hs", c_f), ("cs", c_f), ("hdrop", c_f),
                ("y", c_f), ("y_stride_t", c_i64), ("y_stride_b", c_i64),
                ("persist_ws", c_f), ("persist_ws_bytes", c_i64)]


class LstmSeqGrad(C.Structure):
    _fields_ = [("dy", c_f), ("dy_stride_t", c_i64), ("dy_stride_b", c_i64),
                ("dhT", c_f), ("dcT", c_f),
                ("dgates", c_f), ("dxl", c_f), ("dh", c_f), ("dc", c_f),
                ("dx", c_f), ("lddx", C.c_int32),
                ("dw_ih", _PTRS), ("dw_hh", _PTRS), ("db_ih", _PTRS), ("db_hh", _PTRS),
                ("dh0", c_f), ("dc0", c_f), ("phase", C.c_int32),
                ("persist_ws", c_f), ("persist_ws_bytes", c_i64), ("wide_ws", c_f), ("wide_ws_bytes", c_i64)]


class DecoderSeq(C.Structure):
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("E", C.c_int32),
                ("values", AttnValues),
                ("xemb", c_f), ("w_attn", c_f), ("b_attn", c_f),
                ("w_ih", _PTRS), ("w_hh", _PTRS), ("b_ih", _PTRS), ("b_hh", _PTRS),
                ("w_hhT", _PTRS), ("w_ihT", _PTRS), ("w_ih0cT", c_f), ("w_attn_hT", c_f),
                ("h0", c_f), ("c0", c_f), ("lens", c_f),
                ("dropout_p", c_fl), ("training", C.c_int32), ("seed", c_u64), ("stream_base", c_u64),
                ("seed_offset", c_f),
                ("scores", c_f), ("attn", c_f), ("ld_attn", C.c_int32),
                ("ctx", c_f), ("gates", c_f), ("hs", c_f), ("cs", c_f), ("hdrop", c_f), ("phase", C.c_int32),
                ("h0_stride_l", c_i64), ("attn_ws", c_f), ("attn_ws_bytes", c_i64),
                ("persist_ws", c_f), ("persist_ws_bytes", c_i64)]


class DecoderDecode(C.Structure):
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("E", C.c_int32), ("V", C.c_int32),
                ("values", AttnValues),
                ("emb_table", c_f), ("w_attn", c_f), ("b_attn", c_f),
                ("w_ih", _PTRS), ("w_hh", _PTRS), ("b_ih", _PTRS), ("b_hh", _PTRS),
                ("w_out", c_f), ("b_out", c_f), ("h0", c_f), ("c0", c_f),
                ("start_id", c_i64), ("strategy", C.c_int32), ("seed", c_u64),
                ("target", c_f), ("row_weight", c_f),
                ("ids", c_f), ("loss_rows", c_f), ("attn", c_f), ("ld_attn", C.c_int32),
                ("xemb", c_f), ("scores", c_f), ("ctx", c_f), ("gates", c_f), ("hs", c_f), ("cs", c_f),
                ("logits", c_f), ("keep_logits", C.c_int32)]


class DecoderSeqGrad(C.Structure):
    _fields_ = [("dhtop", c_f), ("dgates", c_f), ("dscores", c_f), ("ld_ds", C.c_int32),
                ("dctx", c_f), ("dh", c_f), ("dc", c_f), ("dxa", c_f), ("dxemb", c_f),
                ("dw_attn", c_f), ("db_attn", c_f),
                ("dw_ih", _PTRS), ("dw_hh", _PTRS), ("db_ih", _PTRS), ("db_hh", _PTRS),
                ("n_text_rows", C.c_int32), ("dtext", c_f), ("dtext_stride_row", c_i64), ("dtext_stride_b", c_i64),
                ("n_video_rows", C.c_int32), ("dvideo", c_f), ("dvideo_stride_row", c_i64), ("dvideo_stride_b", c_i64),
                ("phase", C.c_int32), ("dh_pre", c_f), ("wide_ws", c_f), ("wide_ws_bytes", c_i64),
                ("persist_ws", c_f), ("persist_ws_bytes", c_i64)]


class CopySeg(C.Structure):
    _fields_ = [("dst", c_f), ("src", c_f), ("bytes", c_i64)]


class TransposeJob(C.Structure):
    _fields_ = [("src", c_f), ("ld_src", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("dst", c_f),
                ("ld_dst", C.c_int32)]


class GemmProblem(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("A", c_f), ("lda", C.c_int32), ("B", c_f),
                ("ldb", C.c_int32), ("C", c_f), ("ldc", C.c_int32), ("beta", C.c_int32)]


class BatchPack(C.Structure):
    _fields_ = [("B", C.c_int32), ("Tf", C.c_int32), ("Tc", C.c_int32), ("Td", C.c_int32), ("Da", C.c_int32),
                ("audio_rows", C.c_int32), ("frame_inner", c_i64), ("frames", c_f), ("audio", c_f), ("context", c_f),
                ("target", c_f), ("ctx_len", c_f), ("tgt_len", c_f), ("n_frames", c_f), ("start_id", c_i64),
                ("feats", c_f), ("audio_out", c_f), ("audio_stride_b", c_i64), ("ids_c", c_f), ("ids_d", c_f),
                ("target_t", c_f), ("row_w", c_f), ("ctx_len_out", c_f), ("tgt_len_out", c_f), ("n_frames_out", c_f)]


CNN_MAX_BLOCKS = 4


class CnnBlock(C.Structure):
    _fields_ = [("cout", C.c_int32), ("pool", C.c_int32), ("w", c_f), ("bias", c_f), ("gamma", c_f), ("beta", c_f),
                ("running_mean", c_f), ("running_var", c_f), ("y", c_f), ("z", c_f), ("argmax", c_f), ("stats", c_f),
                ("mean", c_f), ("invstd", c_f), ("scale", c_f), ("shift", c_f)]


class FrameCnn(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("Cin", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("n_blocks", C.c_int32), ("training", C.c_int32), ("time_major", C.c_int32), ("eps", c_fl), ("momentum", c_fl),
                ("frames", c_f), ("n_frames", c_f), ("block", CnnBlock * CNN_MAX_BLOCKS)]


_CPTRS = c_f * CNN_MAX_BLOCKS


class FrameCnnGrad(C.Structure):
    _fields_ = [("dfeat", c_f), ("dconv", c_f), ("dz", c_f), ("dw", _CPTRS), ("dbias", _CPTRS), ("dgamma", _CPTRS),
                ("dbeta", _CPTRS)]


# name -> argtypes (return type is int unless noted); kept in one table so the CPU test can
# check that the shared object exports every symbol the header declares.
SIGNATURES = {
    "mmqg_abi_version": [],
    "mmqg_last_error": [],
    "mmqg_gemm_f32": [c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_i,
                      c_i, c_f],
    "mmqg_gemm_f32_grouped": [c_i, c_i, C.POINTER(GemmProblem), c_i, c_f],
    "mmqg_pack_batch": [C.POINTER(BatchPack), c_f],
    "mmqg_fetch_mapped": [C.POINTER(CopySeg), c_i, c_f],
    "mmqg_embedding_fwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "mmqg_embedding_bwd": [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_f],
    "mmqg_attn_softmax_context_fwd": [C.POINTER(AttnValues), c_f, c_i, c_f, c_i, c_f, c_i, c_f],
    "mmqg_attn_fused_ws_bytes": [C.POINTER(AttnValues), c_i],
    "mmqg_attn_scores_softmax_context_fwd": [C.POINTER(AttnValues), c_f, c_i, c_f, c_i, c_f, c_i, c_i, c_f, c_i, c_f, c_i,
                                             c_f, c_i64, c_f],
    "mmqg_attn_context_bwd": [C.POINTER(AttnValues), c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f],
    "mmqg_attn_context_bwd_fused": [C.POINTER(AttnValues), c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f],
    "mmqg_attn_dvalues": [c_i, c_i, c_i, c_i, c_f, c_i64, c_i, c_i, c_f, c_i64, c_i, c_i, c_f, c_i64, c_i64, c_i, c_f],
    "mmqg_lstm_cell_fwd": [c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_i, c_fl, c_u64, c_u64, c_f],
    "mmqg_lstm_cell_bwd": [c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_i64, c_fl, c_u64, c_u64, c_f, c_i64, c_f, c_f, c_i,
                           c_f, c_i, c_f],
    "mmqg_dropout_mask": [c_f, c_i64, c_fl, c_u64, c_u64, c_f, c_f],
    "mmqg_ce_fwd_bwd": [c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_i, c_f],
    "mmqg_ce_fwd_bwd_stats": [c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_i, c_f],
    "mmqg_projection_stats_ws_bytes": [c_i, c_i],
    "mmqg_projection_last_kernel": [],
    "mmqg_projection_fwd": [c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_f, c_i, c_f, c_i64, C.POINTER(C.c_int32), c_f],
    "mmqg_linear_wgrad": [c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_f],
    "mmqg_colsum_add": [c_f, c_i, c_i, c_i, c_f, c_f],
    "mmqg_reduce_sum": [c_f, c_i, c_f, c_f],
    "mmqg_adam_step": [c_f, c_f, c_f, c_f, c_i64, C.c_double, C.c_double, C.c_double, C.c_double, c_f, c_fl, c_f],
    "mmqg_adam_step_guarded": [c_f, c_f, c_f, c_f, c_i64, C.c_double, C.c_double, C.c_double, C.c_double, c_f, c_fl, c_f, c_f],
    "mmqg_persist_guard_refresh": [c_f, c_f],
    "mmqg_counter_add": [c_f, c_i, c_f],
    "mmqg_transpose_f32": [c_f, c_i, c_i, c_i, c_f, c_i, c_f],
    "mmqg_transpose_f32_batch": [C.POINTER(TransposeJob), c_i, c_f],
    "mmqg_lstm_seq_fwd": [C.POINTER(LstmSeq), c_f],
    "mmqg_lstm_seq_persist_ws_bytes": [C.c_int, C.c_int, C.c_int, C.c_int],
    "mmqg_persist_launch_count": [],
    "mmqg_wide_ws_bytes": [C.c_int, C.c_int],
    "mmqg_lstm_seq_bwd_persist_ws_bytes": [C.c_int, C.c_int, C.c_int, C.c_int],
    "mmqg_persist_bwd_launch_count": [],
    "mmqg_persist_bwd_set_trace": [c_f, c_i64],
    "mmqg_decoder_seq_persist_ws_bytes": [C.POINTER(DecoderSeq)],
    "mmqg_decoder_persist_launch_count": [],
    "mmqg_decoder_persist_set_trace": [c_f, c_i64],
    "mmqg_decoder_seq_bwd_persist_ws_bytes": [C.POINTER(DecoderSeq), C.POINTER(DecoderSeqGrad)],
    "mmqg_decoder_persist_bwd_launch_count": [],
    "mmqg_decoder_persist_bwd_set_trace": [c_f, c_i64],
    "mmqg_persist_declined_count": [],
    "mmqg_persist_failures": [],
    "mmqg_persist_clear_failures": [],
    "mmqg_persist_set_test_fault": [C.c_int, C.c_uint32],
    "mmqg_persist_set_reserved_cus": [C.c_int],
    "mmqg_persist_usable_cus": [c_f, C.c_int],
    "mmqg_persist_set_trace": [c_f, c_i64],
    "mmqg_lstm_seq_bwd": [C.POINTER(LstmSeq), C.POINTER(LstmSeqGrad), c_f],
    "mmqg_lstm_seq_bwd_pair": [C.POINTER(LstmSeq), C.POINTER(LstmSeqGrad), C.POINTER(LstmSeq), C.POINTER(LstmSeqGrad), c_f],
    "mmqg_decoder_decode_run": [C.POINTER(DecoderDecode), c_f],
    "mmqg_sample_gumbel": [c_f, c_i, c_i, c_i, c_u64, c_u64, c_f, c_f],
    "mmqg_frame_cnn_fwd": [C.POINTER(FrameCnn), c_f],
    "mmqg_frame_cnn_bwd": [C.POINTER(FrameCnn), C.POINTER(FrameCnnGrad), c_f],
    "mmqg_decoder_seq_fwd": [C.POINTER(DecoderSeq), c_f],
    "mmqg_decoder_seq_bwd": [C.POINTER(DecoderSeq), C.POINTER(DecoderSeqGrad), c_f],
}

_lib: Optional[C.CDLL] = None


class BackendError(RuntimeError):
    """The HIP extension is missing or a kernel launch was rejected."""


def load() -> C.CDLL:
    """Load the shared object once, bind the signatures and check the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: PyTorch-ROCm ships its own HIP runtime; whichever copy of libamdhip64 is mapped first serves the whole
    # process, and a process in which this extension pulled in /opt/rocm's copy BEFORE torch initialised its own ends up
    # launching through a runtime that sees no device ("no ROCm-capable device is detected" at the first kernel launch)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            f"mmqg: HIP extension not built ({LIB_PATH} missing). Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C multi-modal-qg_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "mmqg_last_error" else (C.c_int64 if name.endswith("_ws_bytes") else C.c_int)
    got = lib.mmqg_abi_version()
    if got != ABI_VERSION:
        raise BackendError(f"mmqg: ABI mismatch: library {got}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mmqg_last_error()
        raise BackendError(f"mmqg {what}: {msg.decode() if msg else 'error'} (rc={rc})")


def ptr(t) -> Optional[int]:
    """Device address of a tensor, or None."""
    return None if t is None else t.data_ptr()
