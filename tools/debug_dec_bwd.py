#!/usr/bin/env python3
"""Debug aid: per-(layer, token) differences of the persistent decoder backward loop against the launched loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import _lib
from mmqg_amd.synthetic import WORKLOADS, Workload, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer

case = sys.argv[1] if len(sys.argv) > 1 else "h128"
if case == "h128":
    w = Workload(case, batch=5, n_frames=4, frame_dim=24, audio_dim=16, ctx_len=7, tgt_len=6, vocab=50, emb_dim=12,
                 hidden=128, layers=3, video_hidden=128, text_max_length=21, av_max_length=9, dropout=0.0)
elif case == "h256":
    w = Workload(case, batch=17, n_frames=5, frame_dim=40, audio_dim=32, ctx_len=9, tgt_len=5, vocab=70, emb_dim=20,
                 hidden=256, layers=3, video_hidden=192, text_max_length=40, av_max_length=12, dropout=0.3)
else:
    c2 = WORKLOADS["config2"]
    w = Workload(**{**c2.dict(), "name": case, "batch": 64, "tgt_len": 4, "vocab": 500})
ragged = os.environ.get("RAGGED", "1") == "1"
vid, text, dec = build_models(w, "cuda", seed=11)
batch = synthetic_batch(w, seed=23, ragged=ragged)
tr = BatchedTrainer(vid, text, dec, batch_size=w.batch, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len, seed=77).train()
lib = _lib.load()
keys = ("dgates_d", "dscores", "dctx", "dh_d", "dc_d")

def run():
    tr.forward_backward(batch)
    torch.cuda.synchronize()
    return {k: tr.ws[k].clone() for k in keys}

got = run()
print("persistent bwd launches", lib.mmqg_decoder_persist_bwd_launch_count(), "tgt_len", batch["tgt_len"].tolist())
pws, pwb = tr.g_dec.persist_ws, tr.g_dec.persist_ws_bytes
tr.g_dec.persist_ws, tr.g_dec.persist_ws_bytes = None, 0
want = run()
T, L = w.tgt_len, 3
for t in range(T - 1, -1, -1):
    line = [f"t={t}"]
    for l in (2, 1, 0):
        a, b = got["dgates_d"][l, t], want["dgates_d"][l, t]
        line.append(f"dG{l} {float((a-b).abs().max()):.2e}/{float(b.abs().max()):.2e}")
    a, b = got["dctx"][t], want["dctx"][t]
    line.append(f"dctx {float((a-b).abs().max()):.2e}/{float(b.abs().max()):.2e}")
    a, b = got["dscores"][t][:, :tr.S], want["dscores"][t][:, :tr.S]
    line.append(f"dS {float((a-b).abs().max()):.2e}/{float(b.abs().max()):.2e}")
    print("  ".join(line))
for k in ("dh_d", "dc_d"):
    for l in range(3):
        a, b = got[k][l], want[k][l]
        print(k, l, f"{float((a-b).abs().max()):.2e}/{float(b.abs().max()):.2e}")
for (l, t) in ((2, T - 1), (1, T - 1), (0, T - 1), (2, T - 2)):
    d = (got["dgates_d"][l, t] - want["dgates_d"][l, t]).abs()
    print(f"dG{l}({t}) err by row:", [f"{x:.1e}" for x in d.max(dim=1).values.tolist()])
    print(f"dG{l}({t}) err by col block of 32:", [f"{x:.1e}" for x in d.view(d.shape[0], -1, 32).amax(dim=(0, 2)).tolist()])
d = (got["dctx"][T - 1] - want["dctx"][T - 1]).abs()
print("dctx(T-1) err by col block of 16:", [f"{x:.1e}" for x in d.view(d.shape[0], -1, 16).amax(dim=(0, 2)).tolist()])
