#!/usr/bin/env python3
"""Where an anti-diagonal of the persistent BACKWARD LSTM time loop spends its time: per-(workgroup, diagonal)
wall-clock stamps (100 MHz) written by the stamped instantiation of the kernel (mmqg_persist_bwd_set_trace).
    python tools/persist_bwd_trace.py [workload]"""
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
tr.forward_backward(b)
torch.cuda.synchronize()
lib = _lib.load()
G, D = 256, tr.Tc + tr.L - 1
buf = torch.zeros(G * D * 6, device="cuda", dtype=torch.int64)
tr.g_text.phase = 1
n0 = lib.mmqg_persist_bwd_launch_count()
for rep in range(3):
    check(lib.mmqg_persist_bwd_set_trace(buf.data_ptr(), buf.numel()))
    check(lib.mmqg_lstm_seq_bwd(C.byref(tr.d_text), C.byref(tr.g_text), ops._stream()))
    check(lib.mmqg_persist_bwd_set_trace(None, 0))
    torch.cuda.synchronize()
tr.g_text.phase = 0
assert lib.mmqg_persist_bwd_launch_count() == n0 + 3, "the persistent backward did not take this shape"
t = buf.view(G, D, 6).cpu().double() * 0.01          # us
start = t[:, :, 0].min(0).values
print(f"text encoder backward, {D} anti-diagonals, {G} workgroups; all times in us")
print(f"whole loop (first start of diagonal 0 -> last barrier exit): {float(t[:, -1, 5].max() - start[0]):.1f}")
per = (t[:, 1:, 0].min(0).values - t[:, :-1, 0].min(0).values)
print(f"diagonal period: mean {float(per.mean()):.2f}  (steady state, diagonals 3..{D - 3}: {float(per[3:-3].mean()):.2f})")
for name, a, b_ in (("phase A: operand loads + MFMA + partial tiles stored (wave 0)", 0, 1),
                    ("first barrier: stores acknowledged, arrival, wait", 1, 2),
                    ("phase B: partial tiles summed, cell backward, dG published", 2, 3),
                    ("exchange stores acknowledged + arrival", 3, 4),
                    ("dG to global memory + wait for the other workgroups", 4, 5)):
    d = (t[:, 3:-3, b_] - t[:, 3:-3, a])
    print(f"{name:66s} mean {float(d.mean()):6.2f}   slowest workgroup per diagonal {float(d.max(0).values.mean()):6.2f}")
for slot, what in ((1, "first"), (4, "second")):
    skew = (t[:, 3:-3, slot].max(0).values - t[:, 3:-3, slot].min(0).values)
    print(f"arrival skew at the {what} barrier (last - first workgroup): mean {float(skew.mean()):.2f}")
