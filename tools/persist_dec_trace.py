#!/usr/bin/env python3
"""Where a token of the persistent decoder FORWARD loop (csrc/persist_dec.hip) spends its time:
per-(workgroup, token) wall-clock stamps (100 MHz) written by the stamped instantiation of the kernel.
    python tools/persist_dec_trace.py [workload]"""
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
G, T = 256, tr.Td
buf = torch.zeros(G * T * 8, device="cuda", dtype=torch.int64)
n0 = lib.mmqg_decoder_persist_launch_count()
tr.d_dec.phase = 2
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    check(lib.mmqg_decoder_persist_set_trace(buf.data_ptr(), buf.numel()))
    e0.record()
    check(lib.mmqg_decoder_seq_fwd(C.byref(tr.d_dec), ops._stream()))
    e1.record()
    check(lib.mmqg_decoder_persist_set_trace(None, 0))
    torch.cuda.synchronize()
tr.d_dec.phase = 0
assert lib.mmqg_decoder_persist_launch_count() == n0 + 3, "the persistent decoder loop did not take this shape"
t = buf.view(G, T, 8).cpu().double() * 0.01          # us
print(f"decoder forward time loop, {T} tokens, {G} workgroups; all times in us; phase-2 call {e0.elapsed_time(e1) * 1e3:.1f}")
print(f"whole loop (first start of token 0 -> last layer-2 arrival): {float(t[:, -1, 7].max() - t[:, 0, 0].min()):.1f}")
per = t[:, 1:, 0].min(0).values - t[:, :-1, 0].min(0).values
print(f"token period: mean {float(per.mean()):.2f}  (tokens 2..{T - 2}: {float(per[2:-2].mean()):.2f})")
names = ("S: score tile (31 workgroups) stored", "barrier after S", "ATT: softmaxes + contexts stored", "barrier after ATT",
         "L0: product + cell + arrive", "L1 (incl. wait for L0's barrier)", "L2 (incl. wait for L1's barrier)")
for i, name in enumerate(names):
    d = t[:, 2:-2, i + 1] - t[:, 2:-2, i]
    print(f"{name:44s} mean {float(d.mean()):6.2f}   slowest workgroup per token {float(d.max(0).values.mean()):6.2f}   fastest {float(d.min(0).values.mean()):6.2f}")
d = t[:, 3:-2, 0] - t[:, 2:-3, 7]
print(f"{'wait for L2 barrier (arrive -> next token)':44s} mean {float(d.mean()):6.2f}")
win = (t[:, 2:-2, 3].max(0).values - t[:, 2:-2, 1].min(0).values)
after = (t[:, 2:-2, 3].max(0).values - t[:, 2:-2, 2].min(0).values)
vals = 4 * tr.B * (tr.Lt * tr.H + tr.Lav * tr.Da + tr.Lav * tr.Dv + 2 * tr.S + tr.Cw)
print(f"attention window (first workgroup past its score tile -> last contexts stored): mean {float(win.mean()):.2f} us = "
      f"{vals / float(win.mean()) / 1e3:.0f} GB/s for the {vals / 1e6:.1f} MB of a token's attention ({vals / float(win.mean()) / 8e6:.3f} of 8 TB/s); "
      f"after the score barrier alone: {float(after.mean()):.2f} us")
