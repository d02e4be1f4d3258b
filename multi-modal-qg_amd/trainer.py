"""Batched training step for the drop-in modules: the MI355X counterpart of the reference's
per-question loop ``train.py:149-181`` (zero_grad -> encoders -> token loops -> summed cross
entropy -> backward -> three Adam steps), for B questions at once.

The three modules keep their reference classes / state-dict keys; this driver

* re-homes all their parameters into ONE flat fp32 buffer (``flat_p``; gradients ``flat_g``,
  Adam moments likewise) so zero_grad is one memset, the optimizer is one fused kernel and the
  data-parallel exchange is a few large RCCL all-reduces instead of one per tensor;
* lays the three attention score layers out back to back (text | audio | video) so the
  per-step scores are one GEMM against a stacked [Lt+2Lav, E+H] matrix;
* keeps the encoder outputs of a batch in one fused value tensor per question
  (text rows | audio rows | video rows) that the attention kernels stream;
* runs forward and backward through the C++ sequence executors (``mmqg_lstm_seq_*``,
  ``mmqg_decoder_seq_*``) — no autograd, no per-token Python — and can capture the whole
  step into a hipGraph.

Loss: ``(1/B) * sum_b sum_{t < tgt_len[b]} CE(logits[b,t], target[b,t])``; for B == 1 this is
train.py:174's running sum.  The embedding table is shared by the text encoder and the decoder
and therefore sits in two of the reference's three Adam optimizers (train.py:236,245,255,
266-267): it is stepped twice per iteration, each time with its own moments.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import K_MAJOR, MN_MAJOR, check, ptr
from .distributed import RESERVED_CUS, GradReducer, broadcast_parameters, exchange_needed, trainer_buckets

_TEXT_STREAM = 1 << 40
_DEC_STREAM = 2 << 40


def _round4(n: int) -> int:
    return (n + 3) // 4 * 4


def _pinned_host_tensor(shape, dtype):
    """Pinned (page-locked, device-mapped) host staging tensor from torch's host allocator, which also owns its lifetime.
    Returns (tensor, device address).  (Registering ordinary tensors with hipHostRegister instead left the registration
    behind when the trainer was collected: the next tensor malloc placed in that address range then failed its
    host-to-device copy with hipErrorInvalidValue.)"""
    t = torch.empty(tuple(shape), dtype=dtype).pin_memory()
    return t, t.data_ptr()


class BatchedTrainer:
    def __init__(self, av_enc_model, text_enc_model, dec_model, *, batch_size: int, n_frames: int, ctx_len: int,
                 tgt_len: int, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, start_id: int = 1,
                 seed: int = 0, mask_mode: Optional[int] = None, process_group=None, use_graph: bool = False,
                 dropout_rank: Optional[int] = None, skip_zero_value_rows: bool = False):
        self.video = getattr(av_enc_model, "video_enc", av_enc_model)
        self.av_model, self.text, self.dec = av_enc_model, text_enc_model, dec_model
        dec = dec_model
        self.dev = dec.out_layer.weight.device
        if self.dev.type != "cuda":
            raise _lib.BackendError("mmqg: BatchedTrainer needs the modules on a ROCm device (model.to('cuda'))")
        _lib.load()
        self.B, self.Tf, self.Tc, self.Td = batch_size, n_frames, ctx_len, tgt_len
        self.L, self.H, self.E, self.V = dec.num_layers, dec.hidden_dim, dec.word_emb_dim, dec.n_vocab
        self.Lt, self.Lav, self.Da, self.Dv = dec.text_max_length, dec.av_max_length, dec.audio_emb_dim, dec.video_emb_dim
        self.Hv, self.Fin = self.video.hidden_dim, self.video.video_emb_dim
        if self.text.num_layers != self.L or self.text.hidden_dim != self.H:
            raise ValueError("text encoder and decoder must share num_layers and hidden_dim (train.py:169 hands the "
                             "encoder state to the decoder)")
        if self.Hv != self.Dv:
            raise ValueError("decoder video_emb_dim must equal the frame encoder's hidden_dim")
        if self.Tc > self.Lt or self.Tf > self.Lav:
            raise ValueError("sequence longer than the attention width")
        if self.text.word_embeddings.weight is not dec.emb_layer.weight:
            raise ValueError("text encoder and decoder must share one embedding layer (train.py:236)")
        self.S = self.Lt + 2 * self.Lav
        self.ldS = _round4(self.S)
        self.Cw = self.H + self.Da + self.Dv
        self.lr, self.betas, self.eps = lr, betas, eps
        self.start_id, self.seed = start_id, seed
        self.mask_mode = dec.mask_mode if mask_mode is None else mask_mode
        # the value rows past a question's context length / frame count are zero padding written by this trainer
        # (train.py:156-160): the attention kernels may skip them with identical results.  Off by default: the
        # benchmark's attention roofline is defined on streaming the padded extents, as the reference's bmm does.
        self.skip_zero_value_rows = bool(skip_zero_value_rows)
        self.pg = process_group
        self.world, rank = 1, 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            rank = torch.distributed.get_rank(process_group)
        # every data-parallel rank draws its dropout masks from streams of its own (bits 48+ of the stream id):
        # with one shared seed, local row b would otherwise get the identical mask on every rank
        self.dropout_rank = rank if dropout_rank is None else int(dropout_rank)
        self.drop_text = float(self.text.dropout_p)
        self.drop_dec = float(dec.dropout_p)
        self.training = True
        self.grad_hook: Optional[Callable[["BatchedTrainer", str], None]] = None
        self._early_adam = False          # set while the single-GPU graph step is warmed up / captured
        self._flatten_parameters()
        self._allocate()
        self._describe()
        # hipGraph replay also for data parallelism: an eager step keeps the host ~90% busy enqueueing, which
        # is fragile with one such process per GPU; the graphs are cut where buckets become final and the
        # all-reduces are issued between them, outside any capture
        self.distributed = exchange_needed(process_group)
        # data parallel: the persistent launch that overlaps collectives leaves CUs to RCCL's channels (distributed.py)
        _lib.load().mmqg_persist_set_reserved_cus(RESERVED_CUS if self.distributed else 0)
        self.use_graph = use_graph
        self._graph = None
        self._cnn_shape, self._cnn_on, self._graph_cnn = None, False, False
        self._tr_jobs = None
        self._audio_rows = 0
        # (stream priorities were tried both ways — side low, chain high — and change nothing: the range on this
        # stack is only (0, -1) and workgroup arbitration between queues does not follow it)
        self._side = self._make_side_stream()
        self._serial = os.environ.get("MMQG_SERIAL", "0") == "1"
        self.chain_first = os.environ.get("MMQG_SIDE_FIRST", "0") != "1"
        if os.environ.get("MMQG_NO_AHEAD", "0") != "1":      # look-ahead recurrent products in the decoder's backward loop
            self.g_dec.dh_pre = self.ws["dpre_d"].data_ptr()
        self.reducer = GradReducer(self.flat_g, trainer_buckets(self.segments, self.n_params), self.pg)
        if self.distributed:
            broadcast_parameters(self.flat_p, self.pg)

    def _make_side_stream(self):
        """The second stream of the step.  MMQG_SIDE_CU_MASK=n (A/B switch) confines it to n of the CUs
        (hipExtStreamCreateWithCUMask; every 256/(256-n)-th CU is left out, so the remainder is spread over all XCDs),
        to see whether the large side-branch GEMMs then stop starving the chain kernels of the main stream."""
        n = int(os.environ.get("MMQG_SIDE_CU_MASK", "0"))
        if n <= 0 or n >= 256:
            return torch.cuda.Stream(device=self.dev)
        hip = C.CDLL("libamdhip64.so")
        drop_every = max(2, round(256 / (256 - n)))
        bits = [0 if (i % drop_every) == drop_every - 1 else 1 for i in range(256)]
        if os.environ.get("MMQG_SIDE_CU_MASK_LOW", "0") == "1":          # alternative: the first n CUs of the enumeration
            bits = [1 if i < n else 0 for i in range(256)]
        words = (C.c_uint32 * 8)(*[sum(b << k for k, b in enumerate(bits[32 * w:32 * w + 32])) for w in range(8)])
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
        if rc != 0:
            raise _lib.BackendError(f"hipExtStreamCreateWithCUMask failed ({rc})")
        self._side_cus = sum(bits)
        return torch.cuda.ExternalStream(st.value, device=self.dev)

    # ------------------------------------------------------------------ parameter layout
    def _flatten_parameters(self):
        dec, text, vid = self.dec, self.text, self.video
        groups: List[Tuple[str, List[torch.nn.Parameter]]] = []
        groups.append(("dec", [dec.text_attn.weight, dec.audio_attn.weight, dec.vid_attn.weight]))   # stacked W_attn
        groups.append(("dec", [dec.text_attn.bias, dec.audio_attn.bias, dec.vid_attn.bias]))          # stacked b_attn
        for p in dec.lstm.flat() + [dec.out_layer.weight, dec.out_layer.bias]:
            groups.append(("dec", [p]))
        seen = {id(p) for _, ps in groups for p in ps}
        emb = dec.emb_layer.weight
        seen.add(id(emb))
        for p in vid.parameters():          # frame encoder right after the decoder: its gradients are final early
            if id(p) not in seen:
                groups.append(("vid", [p]))
                seen.add(id(p))
        for p in text.lstm.flat():
            groups.append(("text", [p]))
        groups.append(("emb", [emb]))
        off = 0
        self.segments: Dict[str, Tuple[int, int]] = {}
        placed = []
        for name, ps in groups:
            off = _round4(off)
            start = off
            for p in ps:
                placed.append((p, off))
                off += p.numel()
            lo, hi = self.segments.get(name, (start, start))
            self.segments[name] = (min(lo, start), off)
        total = _round4(off)
        self.flat_p = torch.zeros(total, device=self.dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=self.dev, dtype=torch.float32)
        self.flat_m = torch.zeros(total, device=self.dev, dtype=torch.float32)
        self.flat_v = torch.zeros(total, device=self.dev, dtype=torch.float32)
        for p, o in placed:
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
        e0, e1 = self.segments["emb"]
        self.emb_m2 = torch.zeros(e1 - e0, device=self.dev, dtype=torch.float32)   # second optimizer's moments
        self.emb_v2 = torch.zeros(e1 - e0, device=self.dev, dtype=torch.float32)
        # two device counters: step_dev = completed steps (mixed into the dropout seed while a step runs, so it may
        # only advance when the step's backward is over); adam_dev = Adam's 1-based step number, advanced before the
        # FIRST optimizer launch of a step (a data-parallel step starts its Adam early, beside the text encoder's backward)
        self.counters = torch.zeros(4, device=self.dev, dtype=torch.int32)
        self.step_dev, self.adam_dev = self.counters[0:1], self.counters[1:2]
        # guard word of the optimizer launches: set on the device (mmqg_persist_guard_refresh) when a persistent time loop
        # of this step has timed out at its barrier — Adam then leaves parameters and moments alone
        self.guard_dev = self.counters[2:3]
        self.n_params = total

    # ------------------------------------------------------------------------ workspaces
    def _allocate(self):
        B, Tf, Tc, Td, L, H, E, V = self.B, self.Tf, self.Tc, self.Td, self.L, self.H, self.E, self.V
        Hv, Fin, Cw, ldS = self.Hv, self.Fin, self.Cw, self.ldS
        dev = self.dev

        def f(*shape):
            return torch.zeros(*shape, device=dev, dtype=torch.float32)

        w = self.ws = {}
        # inputs (static buffers, so a captured graph can be replayed on new data)
        w["feats"] = f(Tf, B, Fin)
        w["ids_c"] = torch.zeros(Tc, B, device=dev, dtype=torch.int64)
        w["ids_d"] = torch.zeros(Td, B, device=dev, dtype=torch.int64)
        w["target"] = torch.zeros(Td, B, device=dev, dtype=torch.int64)
        w["ctx_len"] = torch.zeros(B, device=dev, dtype=torch.int32)
        w["tgt_len"] = torch.zeros(B, device=dev, dtype=torch.int32)
        w["n_frames"] = torch.zeros(B, device=dev, dtype=torch.int32)
        w["row_w"] = f(Td, B)
        # fused value tensor: per question text rows | audio rows | video rows
        self.val_stride = self.Lt * H + self.Lav * self.Da + self.Lav * self.Dv
        w["values"] = f(B, self.val_stride)
        self.off_audio = self.Lt * H
        self.off_video = self.off_audio + self.Lav * self.Da
        # frame LSTM
        w["gates_v"], w["hs_v"], w["cs_v"] = f(1, Tf, B, 4 * Hv), f(1, Tf + 1, B, Hv), f(1, Tf + 1, B, Hv)
        # text encoder
        w["xemb_c"] = f(Tc, B, E)
        w["gates_t"], w["hs_t"], w["cs_t"] = f(L, Tc, B, 4 * H), f(L, Tc + 1, B, H), f(L, Tc + 1, B, H)
        w["hdrop_t"] = f(max(L - 1, 1), Tc, B, H)
        # decoder
        w["xemb_d"] = f(Td, B, E)
        w["scores"], w["attn"], w["ctx"] = f(Td, B, ldS), f(Td, B, ldS), f(Td, B, Cw)
        w["gates_d"], w["hs_d"], w["cs_d"] = f(L, Td, B, 4 * H), f(L, Td + 1, B, H), f(L, Td + 1, B, H)
        w["hdrop_d"] = f(max(L - 1, 1), Td, B, H)
        w["logits"] = f(Td * B, V)
        # loss statistics per row and column tile, written by the projection's epilogue (gemm_nt_tile.hip)
        self._proj_stats_bytes = int(_lib.load().mmqg_projection_stats_ws_bytes(Td * B, V))
        w["proj_stats"] = f(max(self._proj_stats_bytes // 4, 4))
        self._proj_stats_tiles = 0
        w["loss_rows"] = f(Td * B)
        w["argmax"] = torch.zeros(Td * B, device=dev, dtype=torch.int64)
        w["loss"] = f(1)
        # backward
        w["dhtop"] = f(Td, B, H)
        w["dgates_d"], w["dscores"], w["dctx"] = f(L, Td, B, 4 * H), f(Td, B, ldS), f(Td, B, Cw)
        w["dh_d"], w["dc_d"], w["dxa"], w["dpre_d"] = f(L, B, H), f(L, B, H), f(L, B, H), f(L, B, H)
        w["dxemb_d"] = f(Td, B, E)
        w["dtext"], w["dvideo"] = f(Tc, B, H), f(Tf, B, Hv)
        w["dgates_t"], w["dxl_t"], w["dh_t"], w["dc_t"] = f(L, Tc, B, 4 * H), f(Tc, B, H), f(L, B, H), f(L, B, H)
        w["dxemb_c"] = f(Tc, B, E)
        w["dgates_v"], w["dh_v"], w["dc_v"], w["dfeats"] = f(1, Tf, B, 4 * Hv), f(B, Hv), f(B, Hv), f(Tf, B, Fin)
        # k-major (transposed) copies of the recurrent weights for the fused backward time loops
        w["whhT_v"] = f(1, Hv, 4 * Hv)
        w["whhT_t"], w["whhT_d"], w["wihT_d"] = f(L, H, 4 * H), f(L, H, 4 * H), f(L, H, 4 * H)
        w["wihT_t"] = f(L, H, 4 * H)
        w["wih0cT"], w["wattn_hT"] = f(Cw, 4 * H), f(H, ldS)

    # ----------------------------------------------------------------------- descriptors
    def _lstm_ptrs(self, d, params, grads=None):
        for l in range(len(params) // 4):
            wi, wh, bi, bh = params[4 * l:4 * l + 4]
            d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = wi.data_ptr(), wh.data_ptr(), bi.data_ptr(), bh.data_ptr()
            if grads is not None:
                grads.dw_ih[l], grads.dw_hh[l] = wi.grad.data_ptr(), wh.grad.data_ptr()
                grads.db_ih[l], grads.db_hh[l] = bi.grad.data_ptr(), bh.grad.data_ptr()

    def _describe(self):
        w, B, H, L = self.ws, self.B, self.H, self.L
        vals = w["values"]
        # frame LSTM -> video rows of the value tensor
        dv, gv = _lib.LstmSeq(), _lib.LstmSeqGrad()
        dv.T, dv.B, dv.L, dv.H, dv.In = self.Tf, B, 1, self.Hv, self.Fin
        dv.x, dv.ldx = w["feats"].data_ptr(), self.Fin
        self._lstm_ptrs(dv, self.video.lstm.flat(), gv)
        dv.lens = w["n_frames"].data_ptr()
        dv.dropout_p, dv.training, dv.seed, dv.stream_base = 0.0, 1, self.seed, 0
        dv.gates, dv.hs, dv.cs = w["gates_v"].data_ptr(), w["hs_v"].data_ptr(), w["cs_v"].data_ptr()
        dv.y = vals.data_ptr() + 4 * self.off_video
        dv.y_stride_t, dv.y_stride_b = self.Dv, self.val_stride
        gv.dy, gv.dy_stride_t, gv.dy_stride_b = w["dvideo"].data_ptr(), B * self.Hv, self.Hv
        gv.dgates, gv.dh, gv.dc = w["dgates_v"].data_ptr(), w["dh_v"].data_ptr(), w["dc_v"].data_ptr()
        gv.dx, gv.lddx = w["dfeats"].data_ptr(), self.Fin
        dv.w_hhT[0] = w["whhT_v"].data_ptr()
        # (no persistent time loop for the frame LSTM: it runs on the side stream BESIDE the text encoder's, and two
        # persistent launches that cannot share a CU may each end up half resident and wait for the other forever;
        # the library enforces that — csrc/persist_rt.hip declines the second request of a capture's other stream —
        # so asking would only cost the declined attempt)
        self.d_vid, self.g_vid = dv, gv
        # text encoder -> text rows of the value tensor
        dt, gt = _lib.LstmSeq(), _lib.LstmSeqGrad()
        dt.T, dt.B, dt.L, dt.H, dt.In = self.Tc, B, L, H, self.E
        dt.x, dt.ldx = w["xemb_c"].data_ptr(), self.E
        self._lstm_ptrs(dt, self.text.lstm.flat(), gt)
        dt.lens = w["ctx_len"].data_ptr()
        dt.dropout_p, dt.seed, dt.stream_base = self.drop_text, self.seed, _TEXT_STREAM + (self.dropout_rank << 48)
        dt.seed_offset = self.step_dev.data_ptr()
        dt.gates, dt.hs, dt.cs, dt.hdrop = (w[k].data_ptr() for k in ("gates_t", "hs_t", "cs_t", "hdrop_t"))
        dt.y, dt.y_stride_t, dt.y_stride_b = vals.data_ptr(), H, self.val_stride
        gt.dy, gt.dy_stride_t, gt.dy_stride_b = w["dtext"].data_ptr(), B * H, H
        gt.dhT, gt.dcT = w["dh_d"].data_ptr(), w["dc_d"].data_ptr()      # gradient of the state handed to the decoder
        gt.dgates, gt.dxl, gt.dh, gt.dc = (w[k].data_ptr() for k in ("dgates_t", "dxl_t", "dh_t", "dc_t"))
        gt.dx, gt.lddx = w["dxemb_c"].data_ptr(), self.E
        for l in range(L):
            dt.w_hhT[l] = w["whhT_t"][l].data_ptr()
            if l > 0:
                dt.w_ihT[l] = w["wihT_t"][l].data_ptr()
        self._persist_ws(dt, "pws_t")
        n = int(_lib.load().mmqg_lstm_seq_bwd_persist_ws_bytes(dt.T, dt.B, dt.L, dt.H))
        if n > 0:         # the backward time loop as one persistent launch too (csrc/persist_bwd.hip)
            w["pws_tb"] = torch.zeros((n + 3) // 4, device=self.dev, dtype=torch.float32)
            gt.persist_ws, gt.persist_ws_bytes = w["pws_tb"].data_ptr(), n
        self.d_text, self.g_text = dt, gt
        # batches over 64 rows (config 5): workspaces of the wide backward layer-step kernel, one per stack (the frame
        # encoder's backward runs beside the text encoder's)
        for key, grad, width in (("wide_t", gt, H), ("wide_v", gv, self.Hv)):
            n = int(_lib.load().mmqg_wide_ws_bytes(B, width))
            if n > 0:
                w[key] = torch.zeros((n + 3) // 4, device=self.dev, dtype=torch.float32)
                grad.wide_ws, grad.wide_ws_bytes = w[key].data_ptr(), n
        # decoder
        dd, gd = _lib.DecoderSeq(), _lib.DecoderSeqGrad()
        dd.T, dd.B, dd.L, dd.H, dd.E = self.Td, B, L, H, self.E
        v = dd.values
        v.B, v.Lt, v.Lav, v.H, v.Da, v.Dv = B, self.Lt, self.Lav, H, self.Da, self.Dv
        v.text, v.audio, v.video = vals.data_ptr(), vals.data_ptr() + 4 * self.off_audio, vals.data_ptr() + 4 * self.off_video
        v.text_stride_b = v.audio_stride_b = v.video_stride_b = self.val_stride
        v.text_len, v.av_len, v.mask_mode = w["ctx_len"].data_ptr(), w["n_frames"].data_ptr(), self.mask_mode
        v.zero_past_len = int(self.skip_zero_value_rows)
        dd.xemb = w["xemb_d"].data_ptr()
        dec = self.dec
        dd.w_attn, dd.b_attn = dec.text_attn.weight.data_ptr(), dec.text_attn.bias.data_ptr()
        self._lstm_ptrs(dd, dec.lstm.flat(), gd)
        dd.lens = w["tgt_len"].data_ptr()
        dd.dropout_p, dd.seed, dd.stream_base = self.drop_dec, self.seed, _DEC_STREAM + (self.dropout_rank << 48)
        dd.seed_offset = self.step_dev.data_ptr()
        dd.scores, dd.attn, dd.ld_attn, dd.ctx = w["scores"].data_ptr(), w["attn"].data_ptr(), self.ldS, w["ctx"].data_ptr()
        dd.gates, dd.hs, dd.cs, dd.hdrop = (w[k].data_ptr() for k in ("gates_d", "hs_d", "cs_d", "hdrop_d"))
        # the encoder's final (h,c) for every layer, gathered into [L,B,H] (train.py:169)
        w["h0_d"], w["c0_d"] = torch.zeros(L, B, H, device=self.dev), torch.zeros(L, B, H, device=self.dev)
        # training reads the initial state straight from the text encoder's final slot (no gather copy);
        # h0_d / c0_d remain the inputs of the free-running decode
        slot = 4 * self.Tc * B * H
        dd.h0, dd.c0 = w["hs_t"].data_ptr() + slot, w["cs_t"].data_ptr() + slot
        dd.h0_stride_l = (self.Tc + 1) * B * H
        gd.dhtop, gd.dgates = w["dhtop"].data_ptr(), w["dgates_d"].data_ptr()
        gd.dscores, gd.ld_ds, gd.dctx = w["dscores"].data_ptr(), self.ldS, w["dctx"].data_ptr()
        gd.dh, gd.dc, gd.dxa, gd.dxemb = (w[k].data_ptr() for k in ("dh_d", "dc_d", "dxa", "dxemb_d"))
        gd.dw_attn, gd.db_attn = dec.text_attn.weight.grad.data_ptr(), dec.text_attn.bias.grad.data_ptr()
        gd.n_text_rows, gd.dtext = self.Tc, w["dtext"].data_ptr()
        gd.dtext_stride_row, gd.dtext_stride_b = B * H, H
        gd.n_video_rows, gd.dvideo = self.Tf, w["dvideo"].data_ptr()
        gd.dvideo_stride_row, gd.dvideo_stride_b = B * self.Hv, self.Hv
        for l in range(L):
            dd.w_hhT[l] = w["whhT_d"][l].data_ptr()
            if l > 0:
                dd.w_ihT[l] = w["wihT_d"][l].data_ptr()
        dd.w_ih0cT, dd.w_attn_hT = w["wih0cT"].data_ptr(), w["wattn_hT"].data_ptr()
        n = int(_lib.load().mmqg_wide_ws_bytes(B, max(H, self.Cw, self.ldS)))
        if n > 0:
            w["wide_d"] = torch.zeros((n + 3) // 4, device=self.dev, dtype=torch.float32)
            gd.wide_ws, gd.wide_ws_bytes = w["wide_d"].data_ptr(), n
        # score product + softmax + contexts of a step as ONE launch (csrc/attention_fused.hip): opt-in.  Measured at
        # config 2 (round 3): 19.2 us per launch in isolation against 15.7 + 11.1 us for the two launches it replaces,
        # but 30.5 us inside the step, where every token's 54 MB value stream has flushed the score matrix from the L2s
        # and 64 questions re-read it — decoder forward 1.24 ms against 1.17 ms
        n = int(_lib.load().mmqg_attn_fused_ws_bytes(C.byref(dd.values), H)) if os.environ.get("MMQG_ATTN_FUSE", "0") == "1" else 0
        if n > 0:
            w["attn_ws"] = torch.zeros((n + 3) // 4, device=self.dev, dtype=torch.float32)
            dd.attn_ws, dd.attn_ws_bytes = w["attn_ws"].data_ptr(), n
        # the decoder's whole forward time loop as ONE persistent launch (csrc/persist_dec.hip) when the library
        # takes the shape (0 bytes otherwise, or with MMQG_NO_PERSIST_DEC=1)
        n = int(_lib.load().mmqg_decoder_seq_persist_ws_bytes(C.byref(dd)))
        if n > 0:
            w["dec_pws"] = torch.zeros((n + 3) // 4, device=self.dev, dtype=torch.float32)
            dd.persist_ws, dd.persist_ws_bytes = w["dec_pws"].data_ptr(), n
        # ... and its whole BACKWARD time loop (csrc/persist_dec_bwd.hip; round 4): per-token exchange slots, 0.3-0.6 GB
        n = int(_lib.load().mmqg_decoder_seq_bwd_persist_ws_bytes(C.byref(dd), C.byref(gd)))
        if n > 0:
            w["dec_pws_b"] = torch.zeros((n + 3) // 4, device=self.dev, dtype=torch.float32)
            gd.persist_ws, gd.persist_ws_bytes = w["dec_pws_b"].data_ptr(), n
        self.d_dec, self.g_dec = dd, gd

    def _persist_ws(self, d, key):
        """Workspace of the persistent forward time loop (exchange buffers + barrier words), when the
        library takes this shape (csrc/persist.hip)."""
        n = int(_lib.load().mmqg_lstm_seq_persist_ws_bytes(d.T, d.B, d.L, d.H))
        if n > 0:
            self.ws[key] = torch.zeros((n + 3) // 4, device=self.dev, dtype=torch.float32)
            d.persist_ws, d.persist_ws_bytes = self.ws[key].data_ptr(), n

    def set_seed(self, seed: int) -> None:
        """Seed of the dropout streams (mixed with the device-side step counter at run time)."""
        self.seed = int(seed)
        for d in (self.d_vid, self.d_text, self.d_dec):
            d.seed = self.seed
        self._graph = None            # kernel arguments are frozen into captured graphs

    # ------------------------------------------------------------------------- frame CNN
    def _ensure_cnn(self, chw: Tuple[int, int, int]) -> None:
        """Static buffers + descriptor of the HIP frame CNN (encoder.py:40-50,64-67) for raw
        (C,H,W) frames, time-major so that the last block writes straight into the frame LSTM's
        input ``feats`` and reads its gradient from ``dfeats``."""
        if self._cnn_shape == chw:
            return
        vid, B, Tf, dev = self.video, self.B, self.Tf, self.dev
        if vid.kernel_sz != 3 or vid.stride != 1:
            raise _lib.BackendError("mmqg: the batched trainer's frame CNN covers the reference's 3x3 / stride-1 "
                                    "convolutions (config.py:66-67)")
        Cin, Hh, Ww = chw
        N = B * Tf
        d = _lib.FrameCnn(B=B, T=Tf, Cin=Cin, H=Hh, W=Ww, n_blocks=4, training=1, time_major=1,
                          eps=float(vid.bn1.eps), momentum=float(vid.bn1.momentum))
        g = _lib.FrameCnnGrad()
        w = self.ws
        w["raw"] = torch.zeros(Tf, B, Cin, Hh, Ww, device=dev)
        d.frames, d.n_frames = w["raw"].data_ptr(), w["n_frames"].data_ptr()
        keep, h, wd, cin, ymax, xmax = [], Hh, Ww, Cin, 1, 1
        for i, pool in enumerate((False, True, False, True)):
            conv, bn = getattr(vid, f"conv{i + 1}"), getattr(vid, f"bn{i + 1}")
            cout, ho, wo = conv.out_channels, h - 2, wd - 2
            hz, wz = (ho // 3, wo // 3) if pool else (ho, wo)
            if min(ho, wo, hz, wz) < 1:
                raise ValueError(f"frames of {Hh}x{Ww} are too small for the four-block CNN")
            ymax = max(ymax, N * cout * ho * wo)
            if i > 0:
                xmax = max(xmax, N * cin * h * wd)
            last = i == 3
            if last and cout * hz * wz != self.Fin:
                raise ValueError(f"CNN output width {cout * hz * wz} != frame LSTM input {self.Fin} "
                                 f"(flatten_dim / video_emb_dim, config.py:69)")
            y = torch.empty(N, cout, ho, wo, device=dev)
            z = w["feats"] if last else torch.empty(N, cout, hz, wz, device=dev)
            am = torch.empty(N, cout, hz, wz, device=dev, dtype=torch.uint8) if pool else None
            stats = torch.empty(B, cout, 2, device=dev, dtype=torch.float64)
            small = torch.empty(4, B, cout, device=dev)
            blk = d.block[i]
            blk.cout, blk.pool = cout, int(pool)
            blk.w, blk.bias, blk.gamma, blk.beta = (t.data_ptr() for t in (conv.weight, conv.bias, bn.weight, bn.bias))
            blk.running_mean, blk.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            blk.y, blk.z, blk.argmax, blk.stats = y.data_ptr(), z.data_ptr(), ptr(am), stats.data_ptr()
            blk.mean, blk.invstd, blk.scale, blk.shift = (small[j].data_ptr() for j in range(4))
            g.dw[i], g.dbias[i] = conv.weight.grad.data_ptr(), conv.bias.grad.data_ptr()
            g.dgamma[i], g.dbeta[i] = bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr()
            keep += [y, z, am, stats, small]
            h, wd, cin = hz, wz, cout
        w["cnn_dconv"], w["cnn_dz"] = torch.empty(ymax, device=dev), torch.empty(xmax, device=dev)
        g.dfeat, g.dconv, g.dz = w["dfeats"].data_ptr(), w["cnn_dconv"].data_ptr(), w["cnn_dz"].data_ptr()
        self.d_cnn, self.g_cnn, self._cnn_keep, self._cnn_shape = d, g, keep, chw
        self._graph = None

    def _bn_buffers_moved(self) -> bool:
        vid = self.video
        return any(getattr(vid, f"bn{i}").running_mean.data_ptr() != self.d_cnn.block[i - 1].running_mean for i in (1, 2, 3, 4))

    # ------------------------------------------------------------------------------ modes
    def train(self, mode: bool = True):
        self.training = mode
        for m in (self.av_model, self.text, self.dec):
            m.train(mode)
        return self

    def eval(self):
        return self.train(False)

    # --------------------------------------------------------------------------- batches
    def load_batch(self, batch: dict) -> None:
        """Copy one batch into the static input buffers.  ``frames``: (B,Tf,Fin) features or
        (B,Tf,C,H,W) raw frames (already in the reference's viewed layout); ``audio`` (B,Tf,Da);
        ``context`` (B,Tc) ids; ``target`` (B,Td) ids; ``ctx_len``/``tgt_len``/``n_frames`` (B,).
        Raw frames switch the HIP frame CNN on for the following forward / backward."""
        w, B, dev = self.ws, self.B, self.dev
        staged = None
        if batch["frames"].device.type == "cpu" and os.environ.get("MMQG_HOST_BATCH", "dma") != "copy":
            batch, staged = self._stage_host_batch(batch)
        frames = batch["frames"]
        self._cnn_on = frames.dim() == 5
        if self._cnn_on:
            self._ensure_cnn(tuple(frames.shape[2:]))
            if self._bn_buffers_moved():          # load_state_dict keeps storage; .to()/re-assignment does not
                self._cnn_shape = None
                self._ensure_cnn(tuple(frames.shape[2:]))

        def on_dev(t, dtype):
            t = t.to(device=dev, dtype=dtype)
            return t if t.is_contiguous() else t.contiguous()

        frames = on_dev(frames, torch.float32)
        audio = on_dev(batch["audio"], torch.float32)
        ctx, tgt = on_dev(batch["context"], torch.int64), on_dev(batch["target"], torch.int64)
        lens = [on_dev(batch[k], torch.int32) for k in ("ctx_len", "tgt_len", "n_frames")]
        if frames.shape[:2] != (B, self.Tf) or ctx.shape != (B, self.Tc) or tgt.shape != (B, self.Td) or audio.shape[0] != B \
                or audio.shape[1] > self.Lav or audio.shape[2] != self.Da or any(t.shape != (B,) for t in lens):
            raise ValueError("batch does not match the trainer's (batch_size, n_frames, ctx_len, tgt_len) extents")
        if not self._cnn_on and frames.shape[2] != self.Fin:
            raise ValueError(f"frame features are {frames.shape[2]} wide, the frame LSTM expects {self.Fin}")
        # ONE kernel repacks everything into the static time-major inputs (csrc/batch.hip)
        p = _lib.BatchPack(B=B, Tf=self.Tf, Tc=self.Tc, Td=self.Td, Da=self.Da, audio_rows=audio.shape[1],
                           frame_inner=frames[0, 0].numel(), frames=frames.data_ptr(), audio=audio.data_ptr(),
                           context=ctx.data_ptr(), target=tgt.data_ptr(), ctx_len=lens[0].data_ptr(),
                           tgt_len=lens[1].data_ptr(), n_frames=lens[2].data_ptr(), start_id=self.start_id,
                           feats=(w["raw"] if self._cnn_on else w["feats"]).data_ptr(),
                           audio_out=w["values"].data_ptr() + 4 * self.off_audio, audio_stride_b=self.val_stride,
                           ids_c=w["ids_c"].data_ptr(), ids_d=w["ids_d"].data_ptr(), target_t=w["target"].data_ptr(),
                           row_w=w["row_w"].data_ptr(), ctx_len_out=w["ctx_len"].data_ptr(),
                           tgt_len_out=w["tgt_len"].data_ptr(), n_frames_out=w["n_frames"].data_ptr())
        check(_lib.load().mmqg_pack_batch(C.byref(p), ops._stream()), "pack_batch")
        if staged is not None:            # the device staging set may be refilled once this launch has read it
            staged.record(torch.cuda.current_stream())
        # the pack kernel writes audio rows [0, audio.shape[1]); rows a previous, longer batch left behind are
        # padding now (train.py:156 pads with zeros) and the no-op masks attend every row
        rows = audio.shape[1]
        if rows < self._audio_rows:
            w["values"][:, self.off_audio + rows * self.Da:self.off_audio + self._audio_rows * self.Da].zero_()
        self._audio_rows = rows
        # the gradient of the frame LSTM's input is only needed when a CNN produced that input
        self.g_vid.dx = w["dfeats"].data_ptr() if self._cnn_on else None

    _HOST_KEYS = (("frames", torch.float32), ("audio", torch.float32), ("context", torch.int64), ("target", torch.int64),
                  ("ctx_len", torch.int32), ("tgt_len", torch.int32), ("n_frames", torch.int32))

    def _stage_host_batch(self, batch: dict):
        """A batch in HOST memory (what the reference's DataLoader hands over, train.py:144-162) on its way to the device
        without blocking the host between two replays of the step graph: the tensors go into one of two pinned staging
        sets with ONE plain memcpy each, the transfer runs on a
        second stream — beside the previous step, which the host is a step ahead of — into a device staging set, and the
        compute stream waits for that transfer only.  Returns (device batch, event to record after the pack launch).
        MMQG_HOST_BATCH: ``dma`` (default) = asynchronous copies by the copy engine, which needs no CU and therefore runs
        beside the persistent time loops; ``mapped`` = one kernel reads the staging set over PCIe (``mmqg_fetch_mapped``:
        no copy-engine call at all, but it waits for CUs the persistent loops own); ``copy`` = blocking ``.to()`` copies
        (round 3)."""
        mode = os.environ.get("MMQG_HOST_BATCH", "dma")
        sig = tuple((k, tuple(batch[k].shape)) for k, _ in self._HOST_KEYS)
        hb = getattr(self, "_hb", None)
        if hb is None or hb["sig"] != sig:
            hb = dict(sig=sig, slot=0, stream=torch.cuda.Stream(device=self.dev), sets=[])
            for _ in range(2):
                host, hptr = {}, {}
                for k, dt in self._HOST_KEYS:
                    host[k], hptr[k] = _pinned_host_tensor(batch[k].shape, dt)
                devs = {k: torch.empty(batch[k].shape, dtype=dt, device=self.dev) for k, dt in self._HOST_KEYS}
                hb["sets"].append(dict(host=host, hptr=hptr, dev=devs, ready=None, packed=None, keep=None))
            self._hb = hb
        hb["slot"] ^= 1
        st = hb["sets"][hb["slot"]]
        if st["ready"] is not None:
            st["ready"].synchronize()          # the launch that read this set's host buffers two batches ago has run
        src, sptr = {}, {}
        for k, dt in self._HOST_KEYS:
            t = batch[k]
            if mode == "mapped" and t.dtype == dt and t.is_contiguous() and t.is_pinned():
                # already pinned (DataLoader(pin_memory=True)): the kernel reads it in place.  (The copy engine does not:
                # asynchronous copies straight out of the caller's pinned tensors ran 12 % slower than out of the staging
                # set at config 1 — 3.94 against 3.48 ms per step, two boxes — and no faster anywhere else; the memcpy
                # into the staging set is host time beside the previous step.)
                src[k], sptr[k] = t, t.data_ptr()
            else:
                h = st["host"][k]
                if t.dtype == dt and t.is_contiguous():
                    # one plain memcpy: torch's copy_ fans a 4 MB copy out over every CPU it sees (256 on the GPU box, of
                    # which the process may use 16) and took 12 ms per batch there — round 3's "18 ms per step"
                    C.memmove(h.data_ptr(), t.data_ptr(), t.numel() * t.element_size())
                else:
                    h.copy_(t)
                src[k], sptr[k] = h, st["hptr"][k]
        st["keep"] = src                        # alive until the kernel has read them
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(hb["stream"]):
            if st["packed"] is not None:
                hb["stream"].wait_event(st["packed"])        # the pack launch that read this device set two batches ago
            if mode != "mapped":
                for k, _ in self._HOST_KEYS:
                    st["dev"][k].copy_(src[k], non_blocking=True)
            else:
                segs = (_lib.CopySeg * len(self._HOST_KEYS))()
                for sg, (k, _) in zip(segs, self._HOST_KEYS):
                    sg.dst, sg.src, sg.bytes = st["dev"][k].data_ptr(), sptr[k], src[k].numel() * src[k].element_size()
                check(_lib.load().mmqg_fetch_mapped(segs, len(segs), ops._stream()), "fetch_mapped")
            st["ready"] = torch.cuda.Event()
            st["ready"].record(hb["stream"])
        cur.wait_event(st["ready"])
        st["packed"] = torch.cuda.Event()
        return st["dev"], st["packed"]

    # ------------------------------------------------------------------------ one step
    # Two HIP streams: the recurrent time loops are latency-bound chains of small launches, the
    # hoisted products / weight gradients are large GEMMs with no recurrence — so they run beside
    # each other (fork/join with events, which a hipGraph capture records as parallel branches):
    #   forward : [frame LSTM, decoder hoists, weight transposes] || [text encoder]
    #   backward: [vocab wgrad] || [vocab dgrad -> decoder loop], then
    #             [decoder weight grads, frame LSTM backward] || [text encoder backward]
    def _fork(self):
        if self._serial:
            return contextlib.nullcontext()
        self._side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._side)

    def _join(self):
        if not self._serial:
            torch.cuda.current_stream().wait_stream(self._side)

    # Issue order matters as much as the dependency graph: whoever is enqueued first runs first (host
    # enqueue in eager mode, node order in a captured graph).  So at every fork the kernels of the
    # dependent chain are issued BEFORE the side branch: the fork point is marked with an event, the
    # chain is enqueued, and only then the side stream waits for the mark and receives its work.
    def _mark(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def _fork_from(self, ev):
        if self._serial:         # MMQG_SERIAL=1 (A/B switch): everything on ONE stream, branch work behind the chain
            return contextlib.nullcontext()
        self._side.wait_event(ev)
        return torch.cuda.stream(self._side)

    def _forward(self, training: bool):
        lib, w = _lib.load(), self.ws
        L, B, H, V = self.L, self.B, self.H, self.V
        for d in (self.d_vid, self.d_text, self.d_dec):
            d.training = int(training)
        self.d_text.dropout_p = self.drop_text if training else 0.0
        self.d_dec.dropout_p = self.drop_dec if training else 0.0
        emb = self.dec.emb_layer.weight

        # The k-major weight copies are only read by the BACKWARD loops: they go to the tail of the side branch, behind
        # the frame encoder, and the main stream joins the branch at an event recorded IN FRONT of them — the transposes
        # (96 MB of traffic, 50 us alone) then run beside the decoder's forward loop, a latency-bound chain that leaves the
        # memory system idle, instead of in front of the text encoder.  The backward waits for them (self._tr_done).
        # MEASURED (round 3, config 2, 30 steps): 4.683 ms with the transposes moved, 4.655 ms with them in front — the
        # copies are no cheaper beside the chain than in front of it.  Opt-in A/B switch.
        late_tr = os.environ.get("MMQG_TRANSPOSES_LATE", "0") == "1" and self.chain_first
        self._tr_done = None
        side_ready = []

        def side_branch():
            s = ops._stream()
            if self._cnn_on:
                self.d_cnn.training = int(training)
                check(lib.mmqg_frame_cnn_fwd(C.byref(self.d_cnn), s), "frame_cnn_fwd")
            check(lib.mmqg_lstm_seq_fwd(C.byref(self.d_vid), s), "lstm_seq_fwd(frames)")
            if not hoists_first:
                decoder_hoists(s)
            if late_tr:
                side_ready.append(self._mark())          # what the decoder's forward needs of this branch ends here
                self._refresh_transposes()
                self._tr_done = self._mark()

        def decoder_hoists(s):
            ops.embedding_fwd(emb, w["ids_d"], w["xemb_d"].view(-1, self.E))
            self.d_dec.phase = 1
            check(lib.mmqg_decoder_seq_fwd(C.byref(self.d_dec), s), "decoder_seq_fwd(hoists)")
            # always: a backward may follow an eval-mode forward too (tr.eval(); tr.forward_backward(batch)), and
            # its fused loops read the k-major copies
            if not late_tr:
                self._refresh_transposes()

        # The text encoder's persistent time loop owns every CU while it runs, so whatever the other branch still has
        # queued at that point waits for its end and the decoder loop then waits for the branch.  The decoder's hoisted
        # products and the weight transposes (independent of both encoders) therefore go IN FRONT of the time loop on
        # this stream, beside the frame encoder's short chain on the other.
        hoists_first = os.environ.get("MMQG_HOISTS_LAST", "0") != "1"

        def chain():
            s = ops._stream()
            if hoists_first:
                decoder_hoists(s)
            ops.embedding_fwd(emb, w["ids_c"], w["xemb_c"].view(-1, self.E))
            check(lib.mmqg_lstm_seq_fwd(C.byref(self.d_text), s), "lstm_seq_fwd(text)")

        if self.chain_first:
            mark = self._mark()
            chain()
            with self._fork_from(mark):
                side_branch()
        else:
            with self._fork():
                side_branch()
            chain()
        s = ops._stream()
        if side_ready:
            torch.cuda.current_stream().wait_event(side_ready[0])
        else:
            self._join()
        self.d_dec.phase = 2
        check(lib.mmqg_decoder_seq_fwd(C.byref(self.d_dec), s), "decoder_seq_fwd(loop)")
        self.d_dec.phase = 0
        htop = w["hs_d"][L - 1, 1:].reshape(self.Td * B, H)
        out = self.dec.out_layer
        tiles = C.c_int32(0)
        check(lib.mmqg_projection_fwd(self.Td * B, V, H, htop.data_ptr(), H, out.weight.data_ptr(), H, out.bias.data_ptr(),
                                      w["logits"].data_ptr(), V, w["proj_stats"].data_ptr(), self._proj_stats_bytes,
                                      C.byref(tiles), s), "projection_fwd")
        self._proj_stats_tiles = tiles.value     # a host decision that depends on the shape only: safe under graph capture

    def _loss_and_backward(self, part: str = "all"):
        """part 'all': the whole backward with its fork/join branches.  The distributed graph step
        splits it where gradient buckets become final: 'a' = loss, vocabulary projection, decoder loop,
        then the decoder's weight gradients and the frame encoder's backward (decoder and frame-encoder
        buckets are final); 'b' = text encoder backward (the rest)."""
        lib, w = _lib.load(), self.ws
        L, B, H, V, E = self.L, self.B, self.H, self.V, self.E
        R = self.Td * B
        s = ops._stream()
        logits = w["logits"]
        out = self.dec.out_layer
        htop = w["hs_d"][L - 1, 1:].reshape(R, H)
        demb = self.dec.emb_layer.weight.grad
        if part in ("all", "a", "dec"):
            if self._tr_done is not None:            # the k-major weight copies were refreshed at the tail of the side branch
                torch.cuda.current_stream().wait_event(self._tr_done)
                self._tr_done = None
            self._backward_decoder(lib, w, s, logits, out, htop, R)
        if part == "dec":
            return

        # Single GPU: the frame LSTM's backward time loop rides in the text encoder's persistent backward launch (both
        # need only what the decoder's loop has just produced; csrc/persist_bwd.hip) instead of being a chain of eight
        # launches on the other branch; only its weight gradients stay there.  Data parallel (eager and graph alike) keeps
        # it on the side branch: there the frame-encoder bucket's all-reduce should start as early as possible.
        pair = part == "all" and self.chain_first and bool(self.g_text.persist_ws) and not self.distributed and \
            os.environ.get("MMQG_NO_BWD_PAIR", "0") != "1"
        pair_done = []
        # Single GPU, captured step: Adam rides inside the step graph, one segment at a time as its gradients become final,
        # each beside a weight-gradient GEMM of the other stream — chain: [text encoder's weight gradients] Adam(text
        # encoder) Adam(embedding, both optimizers); side: [decoder's weight gradients] Adam(decoder) [frame encoder's weight
        # gradients] Adam(frame encoder).  Adam is HBM-bound, the GEMMs beside it are matrix-bound and leave CUs empty (160
        # and 184 workgroups of one per CU); in a tail of their own the optimizer launches were 136 us of a 4.0 ms step.
        # (A third stream for the decoder's Adam changes nothing: a replayed graph puts it on the side branch's queue.)
        inline_adam = part == "all" and self._early_adam and self._inline_adam_ok()
        dec_done, loop_done = [], []

        def enc_side(which: str = "both"):   # decoder weight gradients ("s1"), frame encoder backward ("s2")
            s2 = ops._stream()
            late = getattr(self, "_late_vocab", None)
            if late is not None and which in ("both", "s1"):
                self._late_vocab = None
                late()
            if which in ("both", "s1"):
                self.g_dec.phase = 2
                check(lib.mmqg_decoder_seq_bwd(C.byref(self.d_dec), C.byref(self.g_dec), s2), "decoder_seq_bwd(wgrad)")
                self.g_dec.phase = 0
                ops.embedding_bwd(w["dxemb_d"].view(-1, E), w["ids_d"], demb)
                if inline_adam:
                    # the decoder segment's gradients are final (and this branch's share of the embedding's): its Adam
                    # here, beside the text encoder's weight-gradient group on the other stream
                    dec_done.append(self._mark())
                    torch.cuda.current_stream().wait_event(loop_done[0])        # (the guard word is refreshed behind the last persistent launch)
                    self._adam("dec")
                if self.distributed and not torch.cuda.is_current_stream_capturing():
                    # the decoder bucket (everything but the embedding) is final here: start its all-reduce
                    # from the side stream so it runs beside the text encoder's backward
                    self.reducer.reduce("dec")
            if which in ("both", "s2"):
                if pair_done:                # its time loop ran in the text encoder's launch on the other stream
                    torch.cuda.current_stream().wait_event(pair_done[0])
                self.g_vid.phase = 2 if pair_done else 0
                check(lib.mmqg_lstm_seq_bwd(C.byref(self.d_vid), C.byref(self.g_vid), s2), "lstm_seq_bwd(frames)")
                self.g_vid.phase = 0
                if self._cnn_on:
                    check(lib.mmqg_frame_cnn_bwd(C.byref(self.d_cnn), C.byref(self.g_cnn), s2), "frame_cnn_bwd")
                if self.distributed and not torch.cuda.is_current_stream_capturing():
                    self.reducer.reduce("vid")          # frame encoder gradients are final too
            if which == "both" and inline_adam:
                self._adam("vid")
            if which == "both" and self._early_adam and not inline_adam:
                # (MMQG_INLINE_ADAM=0, the round-3 arrangement) single GPU, captured step: the decoder's and the frame
                # encoder's gradients are final here, and nothing later in the step reads their parameters — their Adam
                # runs at the end of this branch, the rest in a graph of its own behind the step
                self._adam("early")

        def enc_chain():             # text encoder backward: the rest of the dependent chain
            if pair:
                check(lib.mmqg_lstm_seq_bwd_pair(C.byref(self.d_text), C.byref(self.g_text), C.byref(self.d_vid),
                                                 C.byref(self.g_vid), s), "lstm_seq_bwd_pair(text + frames, loops)")
                pair_done.append(self._mark())
            else:
                self.g_text.phase = 1
                check(lib.mmqg_lstm_seq_bwd(C.byref(self.d_text), C.byref(self.g_text), s), "lstm_seq_bwd(text, loop)")
            if inline_adam:
                self._adam("guard")         # behind the last persistent launch of the step; every optimizer launch reads it
                loop_done.append(self._mark())
            self.g_text.phase = 2
            check(lib.mmqg_lstm_seq_bwd(C.byref(self.d_text), C.byref(self.g_text), s), "lstm_seq_bwd(text, wgrad)")
            self.g_text.phase = 0
            ops.embedding_bwd(w["dxemb_c"].view(-1, E), w["ids_c"], demb)

        if part == "a":
            enc_side()
        elif part in ("s1", "s2"):
            enc_side(part)
        elif part == "b":
            enc_chain()
        elif self.chain_first:
            mark = self._mark()
            if inline_adam:
                self._adam("begin")
            enc_chain()
            if inline_adam:
                self._adam("text")
            with self._fork_from(mark):
                enc_side()
            if inline_adam:
                torch.cuda.current_stream().wait_event(dec_done[0])       # the embedding gradient has a part from either branch
                self._adam("emb")
            self._join()
            if inline_adam:
                self._adam("end")
        else:
            with self._fork():
                enc_side()
            enc_chain()
            self._join()
        if self.grad_hook and part in ("all", "b"):
            self.grad_hook(self, "all")

    def _backward_decoder(self, lib, w, s, logits, out, htop, R):
        V, H = self.V, self.H
        check(lib.mmqg_ce_fwd_bwd_stats(logits.data_ptr(), V, w["target"].data_ptr(), w["row_w"].data_ptr(), R, V,
                                        w["proj_stats"].data_ptr(), self._proj_stats_tiles, w["loss_rows"].data_ptr(),
                                        w["argmax"].data_ptr(), logits.data_ptr(), V, s), "ce_fwd_bwd")
        # vocabulary projection backward (logits now holds dlogits): weight gradient on the side stream
        def vocab_side():            # weight gradient of the projection; the loss scalar is off the chain too
            check(lib.mmqg_reduce_sum(w["loss_rows"].data_ptr(), R, w["loss"].data_ptr(), ops._stream()), "reduce_sum")
            check(lib.mmqg_linear_wgrad(V, H, R, logits.data_ptr(), V, htop.data_ptr(), H, out.weight.grad.data_ptr(), H,
                                        out.bias.grad.data_ptr(), ops._stream()), "linear_wgrad")

        def vocab_chain():
            ops.gemm(K_MAJOR, MN_MAJOR, R, H, V, logits, V, out.weight, H, w["dhtop"], H)
            self.g_dec.phase = 1
            check(lib.mmqg_decoder_seq_bwd(C.byref(self.d_dec), C.byref(self.g_dec), s), "decoder_seq_bwd(loop)")

        # MMQG_VOCAB_WGRAD=late|first (A/B): the projection's weight gradient behind the decoder's loop (with the other
        # weight-gradient GEMMs) / in front of it on the chain, instead of beside the loop on the other stream
        mode = os.environ.get("MMQG_VOCAB_WGRAD", "beside")
        if mode == "late":
            vocab_chain()
            self._late_vocab = vocab_side
            return
        if mode == "first":
            vocab_side()
            vocab_chain()
            return
        if self.chain_first:
            mark = self._mark()
            vocab_chain()
            with self._fork_from(mark):
                vocab_side()
        else:
            with self._fork():
                vocab_side()
            vocab_chain()
        self._join()

    def _refresh_transposes(self):
        """k-major copies of the recurrent weights for the backward loops, rebuilt every step (ONE launch
        on the side stream) so externally loaded weights are honoured."""
        if self._tr_jobs is None:
            w, H, L, Hv = self.ws, self.H, self.L, self.Hv
            jobs = []

            def tr(src_ptr, ld_src, rows, cols, dst, ld_dst):
                jobs.append((src_ptr, ld_src, rows, cols, dst.data_ptr(), ld_dst))

            tr(self.video.lstm.weight_hh_l0.data_ptr(), Hv, 4 * Hv, Hv, w["whhT_v"][0], 4 * Hv)
            for l in range(L):
                tr(getattr(self.text.lstm, f"weight_hh_l{l}").data_ptr(), H, 4 * H, H, w["whhT_t"][l], 4 * H)
                tr(getattr(self.dec.lstm, f"weight_hh_l{l}").data_ptr(), H, 4 * H, H, w["whhT_d"][l], 4 * H)
                if l > 0:
                    tr(getattr(self.dec.lstm, f"weight_ih_l{l}").data_ptr(), H, 4 * H, H, w["wihT_d"][l], 4 * H)
                    tr(getattr(self.text.lstm, f"weight_ih_l{l}").data_ptr(), H, 4 * H, H, w["wihT_t"][l], 4 * H)
            In0 = self.E + self.Cw
            tr(self.dec.lstm.weight_ih_l0.data_ptr() + 4 * self.E, In0, 4 * H, self.Cw, w["wih0cT"], 4 * H)
            tr(self.dec.text_attn.weight.data_ptr() + 4 * self.E, self.E + H, self.S, H, w["wattn_hT"], self.ldS)
            arr = (_lib.TransposeJob * len(jobs))()
            for a, j in zip(arr, jobs):
                a.src, a.ld_src, a.rows, a.cols, a.dst, a.ld_dst = j
            self._tr_jobs = arr
        check(_lib.load().mmqg_transpose_f32_batch(self._tr_jobs, len(self._tr_jobs), ops._stream()), "transpose_f32_batch")

    def _inline_adam_ok(self) -> bool:
        """The captured single-GPU step may carry its optimizer launches inside the step graph (MMQG_INLINE_ADAM=0: the
        round-3 arrangement, decoder + frame encoder at the end of the side branch and the rest in a graph of its own)."""
        return self.chain_first and not self._serial and os.environ.get("MMQG_INLINE_ADAM", "1") != "0"

    def _adam_split(self) -> int:
        """First element of the 'rest' bucket (0: the flat layout has no dec | vid | rest split)."""
        bk = self.reducer.buckets
        return bk["rest"][0] if "rest" in bk and "vid" in bk else 0

    def _adam(self, part: str = "all"):
        """Adam over the flat buffer.  part 'all': everything (+ the embedding's second optimizer), both counters
        advance.  Data-parallel graph step: 'early' = decoder + frame-encoder segments as soon as their buckets are
        reduced (advances adam_dev only: the running backward still reads step_dev), 'late' = the rest."""
        lib, s = _lib.load(), ops._stream()
        b1, b2 = self.betas
        scale = 1.0 / self.world
        split = self._adam_split()
        if part in ("begin", "guard", "text", "dec", "vid", "emb", "end"):
            # the single-GPU graph step's in-graph optimizer (see _loss_and_backward): 'begin' advances Adam's step
            # number, 'guard' refreshes the guard word, 'text' / 'dec' / 'vid' update the text encoder / decoder / frame
            # encoder, 'emb' the embedding (twice: the reference's two optimizers), 'end' closes the step
            bk = self.reducer.buckets
            d0, v0, v1 = bk["dec"][0], bk["vid"][0], bk["rest"][0]
            e0, e1 = self.segments["emb"]
            guard = self.guard_dev.data_ptr()

            def upd(lo, hi, m=None, v=None, what="adam_step"):
                if hi <= lo:
                    return
                check(lib.mmqg_adam_step_guarded(self.flat_p.data_ptr() + 4 * lo, self.flat_g.data_ptr() + 4 * lo,
                                                 (self.flat_m.data_ptr() + 4 * lo) if m is None else m,
                                                 (self.flat_v.data_ptr() + 4 * lo) if v is None else v, hi - lo,
                                                 self.lr, b1, b2, self.eps, self.adam_dev.data_ptr(), scale, guard, s), what)
            if part == "begin":
                check(lib.mmqg_counter_add(self.adam_dev.data_ptr(), 1, s), "counter_add")
            elif part == "guard":
                check(lib.mmqg_persist_guard_refresh(guard, s), "persist_guard_refresh")
            elif part == "text":
                upd(v1, e0)
            elif part == "dec":
                upd(d0, v0)
            elif part == "vid":
                upd(v0, v1)
            elif part == "emb":
                upd(e0, self.n_params)
                upd(e0, e1, self.emb_m2.data_ptr(), self.emb_v2.data_ptr(), "adam_step(embedding, 2nd optimizer)")
            else:
                check(lib.mmqg_counter_add(self.step_dev.data_ptr(), 1, s), "counter_add")
            return
        if part in ("all", "early"):
            check(lib.mmqg_counter_add(self.adam_dev.data_ptr(), 1, s), "counter_add")
        # one device thread copies the persistent kernels' failure word (pinned host memory) into the guard the optimizer
        # launches read: a step whose gradients were poisoned is dropped instead of destroying parameters and moments
        check(lib.mmqg_persist_guard_refresh(self.guard_dev.data_ptr(), s), "persist_guard_refresh")
        guard = self.guard_dev.data_ptr()
        lo, hi = {"all": (0, self.n_params), "early": (0, split), "late": (split, self.n_params)}[part]
        if hi > lo:
            check(lib.mmqg_adam_step_guarded(self.flat_p.data_ptr() + 4 * lo, self.flat_g.data_ptr() + 4 * lo,
                                             self.flat_m.data_ptr() + 4 * lo, self.flat_v.data_ptr() + 4 * lo, hi - lo,
                                             self.lr, b1, b2, self.eps, self.adam_dev.data_ptr(), scale, guard, s), "adam_step")
        if part in ("all", "late"):
            e0, e1 = self.segments["emb"]
            check(lib.mmqg_adam_step_guarded(self.flat_p.data_ptr() + 4 * e0, self.flat_g.data_ptr() + 4 * e0,
                                             self.emb_m2.data_ptr(), self.emb_v2.data_ptr(), e1 - e0, self.lr, b1, b2,
                                             self.eps, self.adam_dev.data_ptr(), scale, guard, s),
                  "adam_step(embedding, 2nd optimizer)")
            check(lib.mmqg_counter_add(self.step_dev.data_ptr(), 1, s), "counter_add")

    def _allreduce(self):
        if self.distributed:
            self.reducer.reduce_remaining()
            self.reducer.finish()

    def forward_backward(self, batch: Optional[dict] = None):
        """zero_grad + forward + loss + backward for the batch already loaded (or ``batch``).
        Gradients are left in ``flat_g`` / every ``param.grad``; returns the loss tensor."""
        if batch is not None:
            self.load_batch(batch)
        self.flat_g.zero_()
        self._forward(self.training)
        self._loss_and_backward()
        self._count_bn_batches()
        return self.ws["loss"]

    def _count_bn_batches(self):
        if self._cnn_on and self.training:                      # BatchNorm2d.num_batches_tracked: one per question
            torch._foreach_add_([getattr(self.video, f"bn{i}").num_batches_tracked for i in (1, 2, 3, 4)], self.B)

    def check_health(self, sync: bool = False) -> None:
        """Raise if a persistent time loop of this process timed out at its device-wide barrier (its workgroups were
        not all resident).  The kernels report that through a word in pinned host memory, so this costs a host read;
        ``sync=True`` first waits for the device (use it before checkpointing or trusting a loss)."""
        if sync:
            torch.cuda.synchronize(self.dev)
        lib = _lib.load()
        if lib.mmqg_persist_failures() > 0:
            raise _lib.BackendError(
                "mmqg: a persistent time loop timed out at a device-wide barrier (another persistent launch or another "
                "process on this device kept part of its grid off the CUs); its outputs were poisoned with NaN and the "
                "parameters may have been updated from them — restore the last checkpoint.  MMQG_NO_PERSIST=1 selects "
                "the launch-per-diagonal time loops.")

    def loss_value(self) -> float:
        """The last step's loss as a float: waits for the device, then checks the health word and finiteness."""
        v = float(self.ws["loss"])
        self.check_health()
        if v != v:
            raise _lib.BackendError("mmqg: the loss is NaN")
        return v

    def step(self, batch: Optional[dict] = None):
        """One full training iteration (train.py:149-181).  Returns the loss tensor (device)."""
        self.check_health()         # a failure of an EARLIER step (the word is written asynchronously): stop before the next update
        if self.use_graph and batch is not None:
            return self._graph_step(batch)
        loss = self.forward_backward(batch)
        self._allreduce()
        self._adam()
        return loss

    # ------------------------------------------------------------------------ hipGraph
    def _graph_body(self, part: str = "all"):
        if part in ("all", "dec"):
            # (clearing the gradient buffer at the head of the side branch instead — 15 us off this chain — was tried in round 4:
            # the frame encoder's chain on that branch then loses its race with the text encoder's persistent launch, its last
            # two steps run behind it and the decoder's loop starts no earlier: `git show c4142df`)
            self.flat_g.zero_()
            self._forward(True)
        self._loss_and_backward(part)

    def _graph_step(self, batch):
        self.load_batch(batch)
        if self._graph is not None and self._graph_cnn != self._cnn_on:
            self._graph = None                                  # raw frames <-> features: different launch sequence
        # single GPU: [zero, forward, loss, backward with its two branches] | [Adam].
        # Data parallel: RCCL stays outside every capture, so the step is cut where gradient buckets become final
        # and the two branches of the encoder backward are replayed on two streams:
        #   main : [zero, forward, loss, decoder backward] ............ [text encoder backward] join, all-reduce(rest),
        #                                                                                       wait, [Adam(rest)]
        #   side :    [decoder weight grads] all-reduce(dec) [frame encoder backward] all-reduce(vid) wait [Adam(dec, vid)]
        # i.e. the first two buckets travel, and their Adam runs, beside the text encoder's backward.
        dp = self.distributed
        parts = ("dec", "s1", "s2", "b") if dp else ("all",)
        if self._graph is None:
            self._graph_cnn = self._cnn_on
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            bn_stats = [b for i in (1, 2, 3, 4) for b in (getattr(self.video, f"bn{i}").running_mean,
                                                          getattr(self.video, f"bn{i}").running_var)] if self._cnn_on else []
            saved = [b.clone() for b in bn_stats]
            # single GPU: Adam of the decoder + frame-encoder segments rides inside the step graph (see enc_side)
            split_early = (not dp) and self.grad_hook is None and self._adam_split() > 0 \
                and os.environ.get("MMQG_NO_EARLY_ADAM", "0") != "1"
            self._early_adam = split_early
            adam_state = [t.clone() for t in (self.flat_p, self.flat_m, self.flat_v, self.emb_m2, self.emb_v2,
                                              self.counters)] if split_early else []
            with torch.cuda.stream(warm):      # warm-up outside capture (lazy code-object loads)
                for part in parts:
                    self._graph_body(part)
            torch.cuda.current_stream().wait_stream(warm)
            for b, v in zip(bn_stats, saved):  # the warm-up pass must not count as a training step
                b.copy_(v)
            for t, v in zip((self.flat_p, self.flat_m, self.flat_v, self.emb_m2, self.emb_v2, self.counters), adam_state):
                t.copy_(v)                      # ... nor update parameters (its early Adam ran)
            torch.cuda.synchronize()
            if dp:
                self.reducer.discard()          # the warm-up pass ran eagerly and may have queued reductions
            self._graphs = {}
            pool = None
            for part in parts:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    self._graph_body(part)
                pool = g.pool()
                self._graphs[part] = g
            self._graph = self._graphs[parts[0]]
            self._early_adam = False
            self._graph_early_adam = split_early
            self._graph_inline_adam = split_early and self._inline_adam_ok()
            for part in (("early", "late") if dp else (() if self._graph_inline_adam else (("late",) if split_early else ("all",)))):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    self._adam(part)
                self._graphs["adam_" + part] = g
        G = self._graphs
        if not dp:
            G["all"].replay()
            self._count_bn_batches()
            if not self._graph_inline_adam:         # (inline: every optimizer launch is a node of the step graph)
                G["adam_late" if self._graph_early_adam else "adam_all"].replay()
            return self.ws["loss"]
        main = torch.cuda.current_stream()
        G["dec"].replay()
        self._count_bn_batches()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            G["s1"].replay()
            self.reducer.reduce("dec")
            G["s2"].replay()
            self.reducer.reduce("vid")
            self.reducer.wait("dec", "vid")
            G["adam_early"].replay()
        G["b"].replay()
        main.wait_stream(self._side)            # the embedding gradient has a part from either branch
        self.reducer.reduce_remaining()
        self.reducer.finish()
        G["adam_late"].replay()
        return self.ws["loss"]

    # ------------------------------------------------------------------------- inference
    @torch.no_grad()
    def decode(self, batch: dict, max_len: Optional[int] = None, strategy: str = "greedy", seed: int = 0,
               with_loss: bool = False, keep_logits: bool = False):
        """Free-running decode of validate() / evaluate() (train.py:87-110, evaluate.py:52-103) for a
        batch, eval mode: encoders once, then ``max_len`` decoder steps that feed each picked token
        back in, all on the device.  strategy: 'greedy' (== 'topk' with k=1) or 'sampling'.
        Returns dict(ids (B,max_len) int64, attn (max_len,B,S), loss (scalar tensor, validate()'s
        per-step CE against ``batch['target']`` averaged like the training loss) or None,
        logits (max_len,B,V) or None).  Tokens after <end> are still produced; use
        ``truncate_at_end`` for evaluate.py's stopping rule."""
        T = max_len or self.Td
        self.load_batch(batch)
        lib, w = _lib.load(), self.ws
        L, B, H, V, E = self.L, self.B, self.H, self.V, self.E
        s = ops._stream()
        for d in (self.d_vid, self.d_text):
            d.training = 0
        if self._cnn_on:
            self.d_cnn.training = 0
            check(lib.mmqg_frame_cnn_fwd(C.byref(self.d_cnn), s), "frame_cnn_fwd")
        self.d_text.dropout_p = 0.0
        emb = self.dec.emb_layer.weight
        check(lib.mmqg_lstm_seq_fwd(C.byref(self.d_vid), s), "lstm_seq_fwd(frames)")
        ops.embedding_fwd(emb, w["ids_c"], w["xemb_c"].view(-1, E))
        check(lib.mmqg_lstm_seq_fwd(C.byref(self.d_text), s), "lstm_seq_fwd(text)")
        w["h0_d"].copy_(w["hs_t"][:, self.Tc])
        w["c0_d"].copy_(w["cs_t"][:, self.Tc])
        dev = self.dev
        ids = torch.zeros(T + 1, B, device=dev, dtype=torch.int64)
        attn = torch.zeros(T, B, self.ldS, device=dev)
        logits = torch.empty((T if keep_logits else 1), B, V, device=dev)
        hs, cs = torch.empty(2, L, B, H, device=dev), torch.empty(2, L, B, H, device=dev)
        xemb, scores, ctx = torch.empty(B, E, device=dev), torch.zeros(B, self.ldS, device=dev), torch.empty(B, self.Cw, device=dev)
        gates = torch.empty(L, B, 4 * H, device=dev)
        loss_rows = target = row_w = None
        if with_loss:
            if T > self.Td:
                raise ValueError("with_loss needs max_len <= the target length the trainer was built for")
            loss_rows = torch.zeros(T, B, device=dev)
            target, row_w = w["target"], w["row_w"]
        d = _lib.DecoderDecode()
        d.T, d.B, d.L, d.H, d.E, d.V = T, B, L, H, E, V
        d.values = self.d_dec.values
        d.emb_table = emb.data_ptr()
        d.w_attn, d.b_attn = self.d_dec.w_attn, self.d_dec.b_attn
        for l in range(L):
            d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = (self.d_dec.w_ih[l], self.d_dec.w_hh[l], self.d_dec.b_ih[l],
                                                          self.d_dec.b_hh[l])
        d.w_out, d.b_out = self.dec.out_layer.weight.data_ptr(), self.dec.out_layer.bias.data_ptr()
        d.h0, d.c0 = w["h0_d"].data_ptr(), w["c0_d"].data_ptr()
        d.start_id = self.start_id
        d.strategy = {"greedy": 0, "topk": 0, "sampling": 1}[strategy]
        d.seed = seed
        d.target, d.row_weight = ptr(target), ptr(row_w)
        d.ids, d.loss_rows = ids.data_ptr(), ptr(loss_rows)
        d.attn, d.ld_attn = attn.data_ptr(), self.ldS
        d.xemb, d.scores, d.ctx, d.gates = xemb.data_ptr(), scores.data_ptr(), ctx.data_ptr(), gates.data_ptr()
        d.hs, d.cs, d.logits, d.keep_logits = hs.data_ptr(), cs.data_ptr(), logits.data_ptr(), int(keep_logits)
        check(lib.mmqg_decoder_decode_run(C.byref(d), s), "decoder_decode_run")
        return dict(ids=ids[1:].t().contiguous(), attn=attn[:, :, :self.S],
                    loss=loss_rows.sum() if with_loss else None, logits=logits if keep_logits else None,
                    # per-question sum of the step losses (train.py:104-106 before the division by target_len)
                    loss_per_question=(loss_rows * B).sum(0) if with_loss else None,
                    hidden=(hs[T % 2], cs[T % 2]))

    @torch.no_grad()
    def logits(self) -> torch.Tensor:
        """(B,Td,V) logits of the last forward (before the loss kernel overwrote them with
        their gradient they are only valid after ``forward_only``)."""
        return self.ws["logits"].view(self.Td, self.B, self.V).transpose(0, 1)

    @torch.no_grad()
    def forward_only(self, batch: dict, training: bool = False) -> torch.Tensor:
        self.load_batch(batch)
        self._forward(training)
        return self.logits()
