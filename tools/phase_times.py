#!/usr/bin/env python3
"""Per-phase GPU time of one training step (each phase captured into its own hipGraph and
replayed), to see where a step's milliseconds go.  python tools/phase_times.py [workload]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import _lib, ops
from mmqg_amd._lib import check
from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "config2"]
vid, text, dec = build_models(w, "cuda", seed=0)
tr = BatchedTrainer(vid, text, dec, batch_size=w.batch, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len).train()
b = {k: v.cuda() for k, v in synthetic_batch(w, seed=0).items()}
tr.step(b)
tr.load_batch(b)
lib = _lib.load()
ws = tr.ws
emb = tr.dec.emb_layer.weight
L, B, H, V, E, Td = tr.L, tr.B, tr.H, tr.V, tr.E, tr.Td
R = Td * B


def s():
    return ops._stream()


def vocab_fwd():
    htop = ws["hs_d"][L - 1, 1:].reshape(R, H)
    tiles = C.c_int32(0)
    out = tr.dec.out_layer
    check(lib.mmqg_projection_fwd(R, V, H, htop.data_ptr(), H, out.weight.data_ptr(), H, out.bias.data_ptr(),
                                  ws["logits"].data_ptr(), V, ws["proj_stats"].data_ptr(), tr._proj_stats_bytes,
                                  C.byref(tiles), s()))
    tr._proj_stats_tiles = tiles.value


def ce():
    check(lib.mmqg_ce_fwd_bwd_stats(ws["logits"].data_ptr(), V, ws["target"].data_ptr(), ws["row_w"].data_ptr(), R, V,
                                    ws["proj_stats"].data_ptr(), tr._proj_stats_tiles, ws["loss_rows"].data_ptr(),
                                    ws["argmax"].data_ptr(), ws["logits"].data_ptr(), V, s()))


def vocab_bwd():
    out = tr.dec.out_layer
    htop = ws["hs_d"][L - 1, 1:].reshape(R, H)
    ops.gemm(0, 1, R, H, V, ws["logits"], V, out.weight, H, ws["dhtop"], H)
    check(lib.mmqg_linear_wgrad(V, H, R, ws["logits"].data_ptr(), V, htop.data_ptr(), H, out.weight.grad.data_ptr(), H,
                                out.bias.grad.data_ptr(), s()))


phases = [
    ("zero_grad", lambda: tr.flat_g.zero_()),
    ("frame lstm fwd", lambda: check(lib.mmqg_lstm_seq_fwd(C.byref(tr.d_vid), s()))),
    ("text emb+lstm fwd", lambda: (ops.embedding_fwd(emb, ws["ids_c"], ws["xemb_c"].view(-1, E)),
                                   check(lib.mmqg_lstm_seq_fwd(C.byref(tr.d_text), s())))),
    ("decoder fwd", lambda: (ops.embedding_fwd(emb, ws["ids_d"], ws["xemb_d"].view(-1, E)),
                             check(lib.mmqg_decoder_seq_fwd(C.byref(tr.d_dec), s())))),
    ("vocab proj fwd", vocab_fwd),
    ("cross entropy", ce),
    ("vocab proj bwd", vocab_bwd),
    ("transposes", tr._refresh_transposes),
    ("decoder bwd", lambda: check(lib.mmqg_decoder_seq_bwd(C.byref(tr.d_dec), C.byref(tr.g_dec), s()))),
    ("text lstm bwd", lambda: check(lib.mmqg_lstm_seq_bwd(C.byref(tr.d_text), C.byref(tr.g_text), s()))),
    ("frame lstm bwd", lambda: check(lib.mmqg_lstm_seq_bwd(C.byref(tr.d_vid), C.byref(tr.g_vid), s()))),
    ("adam", tr._adam),
]
total = 0.0
for name, fn in phases:
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    total += ms
    print(f"{name:22s} {ms*1e3:9.1f} us", flush=True)
print(f"{'sum':22s} {total*1e3:9.1f} us")
