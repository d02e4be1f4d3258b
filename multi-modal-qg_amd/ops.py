"""Tensor-level wrappers over the C ABI and the ``torch.autograd.Function``s that back the
drop-in per-token modules (``model.encoder`` / ``model.decoder``).

PyTorch is plumbing here: it owns device memory and the stream, and autograd only sequences
calls; every arithmetic step below is a hand-written gfx950 kernel reached through
``libmmqg_hip.so``.  CPU tensors are rejected (no fallback).
"""
from __future__ import annotations

import ctypes as C
import itertools
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import K_MAJOR, MN_MAJOR, check, ptr


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.BackendError("mmqg: tensors must live on a ROCm device (cuda:N); the HIP path has no CPU fallback")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    require_device(t)
    if t.dtype != torch.float32:
        raise TypeError(f"mmqg: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------- raw ops
def gemm(a_layout: int, b_layout: int, M: int, N: int, K: int, A, lda: int, B, ldb: int, Cm, ldc: int, *,
         beta: int = 0, bias=None, bias2=None, A2=None, lda2: int = 0, B2=None, ldb2: int = 0, K2: int = 0,
         split_k: int = -1, a_off: int = 0, b_off: int = 0, c_off: int = 0) -> None:
    """C = beta*C + bias + bias2 + op(A)*op(B) [+ op(A2)*op(B2)]; *_off are element offsets."""
    lib = _lib.load()
    pa = A.data_ptr() + 4 * a_off
    pb = B.data_ptr() + 4 * b_off
    pc = Cm.data_ptr() + 4 * c_off
    check(lib.mmqg_gemm_f32(a_layout, b_layout, M, N, K, pa, lda, pb, ldb, ptr(A2), lda2, ptr(B2), ldb2, K2,
                            ptr(bias), ptr(bias2), beta, pc, ldc, split_k, _stream()), "gemm_f32")


def linear_fwd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """y = x w^T + b with torch.nn.Linear layouts: x [M,K], w [N,K]."""
    x, w = _f32c(x), _f32c(w)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    gemm(K_MAJOR, K_MAJOR, M, N, K, x, K, w, K, y, N, bias=None if b is None else _f32c(b))
    return y


def embedding_fwd(table: torch.Tensor, ids: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    table = _f32c(table)
    require_device(ids)
    ids = ids.reshape(-1).to(torch.int64).contiguous()
    V, E = table.shape
    if out is None:
        out = torch.empty(ids.numel(), E, device=table.device, dtype=torch.float32)
    check(_lib.load().mmqg_embedding_fwd(table.data_ptr(), ids.data_ptr(), out.data_ptr(), ids.numel(), V, E,
                                         out.stride(0), _stream()), "embedding_fwd")
    return out


def embedding_bwd(dout: torch.Tensor, ids: torch.Tensor, dtable: torch.Tensor) -> None:
    dout = _f32c(dout)
    ids = ids.reshape(-1).to(torch.int64).contiguous()
    V, E = dtable.shape
    check(_lib.load().mmqg_embedding_bwd(dout.data_ptr(), dout.stride(0), ids.data_ptr(), dtable.data_ptr(),
                                         ids.numel(), V, E, _stream()), "embedding_bwd")


def make_attn_values(text, audio, video, text_len=None, av_len=None, mask_mode: int = 0) -> _lib.AttnValues:
    """text [B,Lt,H], audio [B,Lav,Da], video [B,Lav,Dv]; last two dims contiguous."""
    require_device(text, audio, video, text_len, av_len)
    B, Lt, H = text.shape
    _, Lav, Da = audio.shape
    Dv = video.shape[2]
    for t in (text, audio, video):
        if t.dtype != torch.float32 or t.stride(2) != 1 or t.stride(1) != t.shape[2] or t.shape[0] != B:
            raise ValueError("mmqg: value tensors must be float32 [B,L,D] with contiguous rows")
    if video.shape[1] != Lav:
        raise ValueError("mmqg: audio and video value tensors must have the same number of rows")
    v = _lib.AttnValues()
    v.B, v.Lt, v.Lav, v.H, v.Da, v.Dv = B, Lt, Lav, H, Da, Dv
    v.text, v.text_stride_b = text.data_ptr(), text.stride(0) if B > 1 else Lt * H
    v.audio, v.audio_stride_b = audio.data_ptr(), audio.stride(0) if B > 1 else Lav * Da
    v.video, v.video_stride_b = video.data_ptr(), video.stride(0) if B > 1 else Lav * Dv
    v.text_len = ptr(text_len)
    v.av_len = ptr(av_len)
    v.mask_mode = mask_mode
    return v


def dropout_mask(n: int, p: float, seed: int, stream_id: int, device, seed_offset=None) -> torch.Tensor:
    out = torch.empty(n, device=device, dtype=torch.float32)
    check(_lib.load().mmqg_dropout_mask(out.data_ptr(), n, p, seed, stream_id, ptr(seed_offset), _stream()),
          "dropout_mask")
    return out


def projection_fwd(h: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], want_stats: bool = True,
                   ld: Optional[int] = None):
    """logits = h @ weight.T + bias (decoder.py:106) -> (logits, stats, stats_tiles); stats_tiles == 0 when the
    shape was served by the generic product (no loss statistics written).  ld: row pitch of the logits (>= V)."""
    h, weight = _f32c(h), _f32c(weight)
    rows, H = h.shape
    V = weight.shape[0]
    lib = _lib.load()
    logits = torch.empty(rows, ld or V, device=h.device, dtype=torch.float32)[:, :V]
    nbytes = int(lib.mmqg_projection_stats_ws_bytes(rows, V)) if want_stats else 0
    stats = torch.empty(max(nbytes // 4, 4), device=h.device, dtype=torch.float32)
    tiles = C.c_int32(0)
    check(lib.mmqg_projection_fwd(rows, V, H, h.data_ptr(), h.stride(0), weight.data_ptr(), weight.stride(0), ptr(bias),
                                  logits.data_ptr(), logits.stride(0), stats.data_ptr() if want_stats else None, nbytes,
                                  C.byref(tiles), _stream()), "projection_fwd")
    return logits, stats, tiles.value


def ce_fwd_bwd(logits: torch.Tensor, target: Optional[torch.Tensor], row_weight: Optional[torch.Tensor],
               want_grad: bool, in_place: bool = False, stats: Optional[torch.Tensor] = None, stats_tiles: int = 0):
    """Returns (loss_rows, argmax, dlogits).  stats/stats_tiles: the row statistics projection_fwd wrote."""
    require_device(logits)
    if logits.dtype != torch.float32 or logits.dim() != 2 or logits.stride(1) != 1:
        logits = _f32c(logits)
    rows, V = logits.shape
    loss_rows = torch.empty(rows, device=logits.device, dtype=torch.float32) if target is not None else None
    argmax = torch.empty(rows, device=logits.device, dtype=torch.int64)
    dlogits = None
    if want_grad:
        dlogits = logits if in_place else torch.empty_like(logits)
    check(_lib.load().mmqg_ce_fwd_bwd_stats(logits.data_ptr(), logits.stride(0), ptr(target), ptr(row_weight), rows, V,
                                            ptr(stats) if stats_tiles else None, stats_tiles, ptr(loss_rows),
                                            argmax.data_ptr(), ptr(dlogits),
                                            dlogits.stride(0) if dlogits is not None else 0, _stream()), "ce_fwd_bwd")
    return loss_rows, argmax, dlogits


def linear_wgrad(dY: torch.Tensor, X: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor]) -> None:
    """dW += dY^T X, dbias += dY.sum(0) (the weight / bias gradient of y = x W^T + b, decoder.py:106); rows contiguous."""
    require_device(dY, X, dW, dbias)
    rows, out_f = dY.shape
    in_f = X.shape[1]
    check(_lib.load().mmqg_linear_wgrad(out_f, in_f, rows, dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0),
                                        dW.data_ptr(), dW.stride(0), ptr(dbias), _stream()), "linear_wgrad")


def colsum_add(X: torch.Tensor, out: torch.Tensor) -> None:
    M, N = X.shape
    check(_lib.load().mmqg_colsum_add(X.data_ptr(), X.stride(0), M, N, out.data_ptr(), _stream()), "colsum_add")


# ----------------------------------------------------------------------------- autograd glue
class EmbeddingFn(torch.autograd.Function):
    """nn.Embedding lookup (encoder.py:96, decoder.py:75) — K3."""

    @staticmethod
    def forward(ctx, table, ids):
        ctx.save_for_backward(ids)
        ctx.shape = table.shape
        return embedding_fwd(table, ids)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dtable = torch.zeros(ctx.shape, device=dout.device, dtype=torch.float32)
        embedding_bwd(dout, ids, dtable)
        return dtable, None


class MultiLinearFn(torch.autograd.Function):
    """y = [x W1^T + b1 | x W2^T + b2 | ...]: several Linear layers sharing one input, written
    side by side into one row (the three attention score layers of decoder.py:78,84,92, or a
    single Linear such as out_layer, decoder.py:106)."""

    @staticmethod
    def forward(ctx, x, *wb):
        x = _f32c(x)
        ws = [_f32c(w) for w in wb[0::2]]
        bs = [None if b is None else _f32c(b) for b in wb[1::2]]
        M, K = x.shape
        widths = [w.shape[0] for w in ws]
        S = sum(widths)
        y = torch.empty(M, S, device=x.device, dtype=torch.float32)
        off = 0
        for w, b, n in zip(ws, bs, widths):
            gemm(K_MAJOR, K_MAJOR, M, n, K, x, K, w, K, y, S, bias=b, c_off=off)
            off += n
        ctx.save_for_backward(x, *ws)
        ctx.has_bias = [b is not None for b in bs]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, *ws = ctx.saved_tensors
        dy = _f32c(dy)
        M, K = x.shape
        S = dy.shape[1]
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        grads: List[Optional[torch.Tensor]] = []
        off = 0
        for i, w in enumerate(ws):
            n = w.shape[0]
            if dx is not None:       # dx += dy_seg * W
                gemm(K_MAJOR, MN_MAJOR, M, K, n, dy, S, w, K, dx, K, beta=1, a_off=off)
            dw = db = None
            if ctx.needs_input_grad[1 + 2 * i]:     # dW = dy_seg^T * x
                dw = torch.zeros_like(w)
                gemm(MN_MAJOR, MN_MAJOR, n, K, M, dy, S, x, K, dw, K, beta=1, a_off=off)
            if ctx.has_bias[i] and ctx.needs_input_grad[2 + 2 * i]:
                db = torch.zeros(n, device=x.device, dtype=torch.float32)
                check(_lib.load().mmqg_colsum_add(dy.data_ptr() + 4 * off, S, M, n, db.data_ptr(), _stream()), "colsum_add")
            grads += [dw, db]
            off += n
        return (dx, *grads)


class AttentionFn(torch.autograd.Function):
    """softmax over the three score segments + the three weighted value sums
    (decoder.py:79-81,85-87,93-95) — K1."""

    @staticmethod
    def forward(ctx, scores, text, audio, video, text_len, av_len, mask_mode):
        scores = _f32c(scores)
        text, audio, video = _f32c(text), _f32c(audio), _f32c(video)
        v = make_attn_values(text, audio, video, text_len, av_len, mask_mode)
        B, S = scores.shape
        Cw = v.H + v.Da + v.Dv
        attn = torch.empty_like(scores)
        ctxv = torch.empty(B, Cw, device=scores.device, dtype=torch.float32)
        check(_lib.load().mmqg_attn_softmax_context_fwd(C.byref(v), scores.data_ptr(), S, attn.data_ptr(), S,
                                                        ctxv.data_ptr(), Cw, _stream()), "attn_softmax_context_fwd")
        ctx.save_for_backward(attn, text, audio, video, text_len, av_len)
        ctx.mask_mode = mask_mode
        return attn, ctxv

    @staticmethod
    def backward(ctx, dattn, dctx):
        attn, text, audio, video, text_len, av_len = ctx.saved_tensors
        v = make_attn_values(text, audio, video, text_len, av_len, ctx.mask_mode)
        B, S = attn.shape
        Cw = v.H + v.Da + v.Dv
        dctx = _f32c(dctx) if dctx is not None else torch.zeros(B, Cw, device=attn.device, dtype=torch.float32)
        dattn = None if dattn is None else _f32c(dattn)
        dscores = torch.empty_like(attn)
        lib = _lib.load()
        check(lib.mmqg_attn_context_bwd(C.byref(v), attn.data_ptr(), S, dctx.data_ptr(), Cw, ptr(dattn), S,
                                        dscores.data_ptr(), S, _stream()), "attn_context_bwd")
        outs = []
        for i, (val, seg, off, L, D) in enumerate(((text, 0, 0, v.Lt, v.H), (audio, v.Lt, v.H, v.Lav, v.Da),
                                                   (video, v.Lt + v.Lav, v.H + v.Da, v.Lav, v.Dv))):
            if not ctx.needs_input_grad[1 + i]:
                outs.append(None)
                continue
            dv = torch.empty_like(val)
            check(lib.mmqg_attn_dvalues(1, B, L, D, attn.data_ptr(), 0, S, seg, dctx.data_ptr(), 0, Cw, off,
                                        dv.data_ptr(), D, L * D, 0, _stream()), "attn_dvalues")
            outs.append(dv)
        return (dscores, *outs, None, None, None)


_stream_counter = itertools.count(1)


class LSTMSeqFn(torch.autograd.Function):
    """Stacked LSTM over T steps via the C++ sequence executor (encoder.py:54,91,98;
    decoder.py:69,104).  x [T,B,In]; h0,c0 [L,B,H]; params = w_ih_l0,w_hh_l0,b_ih_l0,b_hh_l0,...
    Returns (y [T,B,H] top-layer outputs, hT [L,B,H], cT [L,B,H])."""

    @staticmethod
    def forward(ctx, x, h0, c0, dropout_p, training, seed, *params):
        x, h0, c0 = _f32c(x), _f32c(h0), _f32c(c0)
        params = [_f32c(p) for p in params]
        T, B, In = x.shape
        L, _, H = h0.shape
        dev = x.device
        d = _lib.LstmSeq()
        d.T, d.B, d.L, d.H, d.In = T, B, L, H, In
        d.x, d.ldx = x.data_ptr(), In
        for l in range(L):
            d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = (p.data_ptr() for p in params[4 * l:4 * l + 4])
        d.h0, d.c0, d.lens = h0.data_ptr(), c0.data_ptr(), None
        drop = bool(training) and dropout_p > 0 and L > 1
        d.dropout_p, d.training, d.seed = float(dropout_p), int(bool(training)), int(seed)
        d.stream_base = next(_stream_counter) << 32
        gates = torch.empty(L, T, B, 4 * H, device=dev, dtype=torch.float32)
        hs = torch.empty(L, T + 1, B, H, device=dev, dtype=torch.float32)
        cs = torch.empty(L, T + 1, B, H, device=dev, dtype=torch.float32)
        hdrop = torch.empty(L - 1, T, B, H, device=dev, dtype=torch.float32) if drop else None
        d.gates, d.hs, d.cs, d.hdrop = gates.data_ptr(), hs.data_ptr(), cs.data_ptr(), ptr(hdrop)
        d.y = None
        nws = int(_lib.load().mmqg_lstm_seq_persist_ws_bytes(T, B, L, H))     # > 0: the persistent time loop takes this shape
        pws = torch.zeros((nws + 3) // 4, device=dev, dtype=torch.float32) if nws > 0 else None
        d.persist_ws, d.persist_ws_bytes = ptr(pws), nws
        check(_lib.load().mmqg_lstm_seq_fwd(C.byref(d), _stream()), "lstm_seq_fwd")
        ctx.desc = d
        ctx.keep = (x, h0, c0, params, gates, hs, cs, hdrop, pws)
        y = hs[L - 1, 1:].clone()
        return y, hs[:, T].clone(), cs[:, T].clone()

    @staticmethod
    def backward(ctx, dy, dhT, dcT):
        d = ctx.desc
        x, h0, c0, params, gates, hs, cs, hdrop, _ = ctx.keep
        T, B, L, H, In = d.T, d.B, d.L, d.H, d.In
        dev = x.device
        g = _lib.LstmSeqGrad()
        dy = None if dy is None else _f32c(dy)
        dhT = None if dhT is None else _f32c(dhT)
        dcT = None if dcT is None else _f32c(dcT)
        g.dy, g.dy_stride_t, g.dy_stride_b = ptr(dy), B * H, H
        g.dhT, g.dcT = ptr(dhT), ptr(dcT)
        dgates = torch.empty(L, T, B, 4 * H, device=dev, dtype=torch.float32)
        dxl = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        dh = torch.empty(L, B, H, device=dev, dtype=torch.float32)
        dc = torch.empty(L, B, H, device=dev, dtype=torch.float32)
        dx = torch.empty(T, B, In, device=dev, dtype=torch.float32)
        dh0 = torch.empty(L, B, H, device=dev, dtype=torch.float32)
        dc0 = torch.empty(L, B, H, device=dev, dtype=torch.float32)
        g.dgates, g.dxl, g.dh, g.dc = dgates.data_ptr(), dxl.data_ptr(), dh.data_ptr(), dc.data_ptr()
        g.dx, g.lddx = dx.data_ptr(), In
        g.dh0, g.dc0 = dh0.data_ptr(), dc0.data_ptr()
        nws = int(_lib.load().mmqg_lstm_seq_bwd_persist_ws_bytes(T, B, L, H))   # > 0: the persistent backward takes this shape
        bws = torch.zeros((nws + 3) // 4, device=dev, dtype=torch.float32) if nws > 0 else None
        g.persist_ws, g.persist_ws_bytes = ptr(bws), nws
        nwide = int(_lib.load().mmqg_wide_ws_bytes(B, H))              # > 0: batch over 64 rows, the wide backward layer-step
        wws = torch.zeros((nwide + 3) // 4, device=dev, dtype=torch.float32) if nwide > 0 else None
        g.wide_ws, g.wide_ws_bytes = ptr(wws), nwide
        dparams = [torch.zeros_like(p) for p in params]
        for l in range(L):
            g.dw_ih[l], g.dw_hh[l], g.db_ih[l], g.db_hh[l] = (p.data_ptr() for p in dparams[4 * l:4 * l + 4])
        check(_lib.load().mmqg_lstm_seq_bwd(C.byref(d), C.byref(g), _stream()), "lstm_seq_bwd")
        return (dx, dh0, dc0, None, None, None, *dparams)


class FrameCNNFn(torch.autograd.Function):
    """The conv3x3 -> ReLU -> per-question BatchNorm (-> 3x3/3 max-pool) blocks of
    VideoConvLstmEncoder (reference encoder.py:40-50,64-67) in the HIP frame-CNN kernels.
    frames (B,T,C,H,W) in the reference's view layout; blocks: [(pool, w, bias, gamma, beta,
    running_mean, running_var)]; returns the last block's output (B*T, C', h', w').  Training
    mode advances the running statistics in place, once per question."""

    @staticmethod
    def forward(ctx, frames, n_frames, training, eps, momentum, pools, *params):
        frames = _f32c(frames)
        B, T, Cin, H, W = frames.shape
        nb = len(pools)
        if nb > _lib.CNN_MAX_BLOCKS or len(params) != 6 * nb:
            raise ValueError("FrameCNNFn: expected 6 tensors per block and at most %d blocks" % _lib.CNN_MAX_BLOCKS)
        dev, N = frames.device, B * T
        nf = None if n_frames is None else n_frames.to(device=dev, dtype=torch.int32).clamp(0, T).contiguous()
        d = _lib.FrameCnn(B=B, T=T, Cin=Cin, H=H, W=W, n_blocks=nb, training=int(bool(training)), eps=float(eps),
                          momentum=float(momentum), frames=ptr(frames), n_frames=ptr(nf))
        keep, h, w, cin = [frames, nf], H, W, Cin
        for i in range(nb):
            wt, bias, gamma, beta, rmean, rvar = (_f32c(t) for t in params[6 * i:6 * i + 6])
            cout = wt.shape[0]
            if tuple(wt.shape) != (cout, cin, 3, 3):
                raise ValueError(f"FrameCNNFn: block {i} weight {tuple(wt.shape)} is not ({cout},{cin},3,3)")
            ho, wo = h - 2, w - 2
            hz, wz = (ho // 3, wo // 3) if pools[i] else (ho, wo)
            if min(ho, wo, hz, wz) < 1:
                raise ValueError("FrameCNNFn: image too small for the CNN")
            y = torch.empty(N, cout, ho, wo, device=dev)
            z = torch.empty(N, cout, hz, wz, device=dev)
            am = torch.empty(N, cout, hz, wz, device=dev, dtype=torch.uint8) if pools[i] else None
            stats = torch.empty(B, cout, 2, device=dev, dtype=torch.float64)
            small = torch.empty(4, B, cout, device=dev)
            blk = d.block[i]
            blk.cout, blk.pool = cout, int(bool(pools[i]))
            blk.w, blk.bias, blk.gamma, blk.beta = ptr(wt), ptr(bias), ptr(gamma), ptr(beta)
            blk.running_mean, blk.running_var = ptr(rmean), ptr(rvar)
            blk.y, blk.z, blk.argmax, blk.stats = ptr(y), ptr(z), ptr(am), ptr(stats)
            blk.mean, blk.invstd, blk.scale, blk.shift = (ptr(small[j]) for j in range(4))
            keep += [wt, bias, gamma, beta, rmean, rvar, y, z, am, stats, small]
            h, w, cin = hz, wz, cout
        check(_lib.load().mmqg_frame_cnn_fwd(C.byref(d), _stream()), "frame_cnn_fwd")
        ctx.desc, ctx.keep, ctx.nb, ctx.training = d, keep, nb, bool(training)
        ctx.shapes = [tuple(params[6 * i].shape) for i in range(nb)]
        return keep[-4].view(N, cin, h, w)

    @staticmethod
    def backward(ctx, dfeat):
        if not ctx.training:
            raise _lib.BackendError("mmqg: the frame CNN's backward is defined for training mode (batch statistics)")
        d, nb = ctx.desc, ctx.nb
        dfeat = _f32c(dfeat)
        dev = dfeat.device
        N = d.B * d.T
        ymax, xmax, h, w, cin = 0, 0, d.H, d.W, d.Cin
        for i in range(nb):
            cout, ho, wo = d.block[i].cout, h - 2, w - 2
            ymax = max(ymax, N * cout * ho * wo)
            if i > 0:
                xmax = max(xmax, N * cin * h * w)
            h, w = (ho // 3, wo // 3) if d.block[i].pool else (ho, wo)
            cin = cout
        dconv = torch.empty(max(ymax, 1), device=dev)
        dz = torch.empty(max(xmax, 1), device=dev)
        g = _lib.FrameCnnGrad(dfeat=ptr(dfeat), dconv=ptr(dconv), dz=ptr(dz))
        grads = []
        for i in range(nb):
            cout = d.block[i].cout
            gw = torch.zeros(ctx.shapes[i], device=dev)
            small = torch.zeros(3, cout, device=dev)
            g.dw[i], g.dbias[i], g.dgamma[i], g.dbeta[i] = ptr(gw), ptr(small[0]), ptr(small[1]), ptr(small[2])
            grads += [gw, small[0], small[1], small[2], None, None]
        check(_lib.load().mmqg_frame_cnn_bwd(C.byref(d), C.byref(g), _stream()), "frame_cnn_bwd")
        return (None, None, None, None, None, None, *grads)


def lstm_seq(x, h0, c0, params: Sequence[torch.Tensor], dropout_p: float, training: bool, seed: int):
    return LSTMSeqFn.apply(x, h0, c0, dropout_p, training, seed, *params)
