#!/usr/bin/env python3
"""Where a token of the persistent decoder BACKWARD loop (csrc/persist_dec_bwd.hip) spends its time:
per-(workgroup, token) wall-clock stamps (100 MHz) written by the stamped instantiation of the kernel.
    python tools/persist_dec_bwd_trace.py [workload]
Stamp slots per (workgroup, token): 0 token start; 1 past the attention barrier of token t+1 (SW starts); 2 SW partial
tiles stored; for the three cell stages P2, P1, P0 (i = 0, 1, 2): 3+3i past the previous barrier, 4+3i cell backward done
(dG fragment in LDS), 5+3i late product stored (arrival follows); 12 past P0's barrier (attention starts), 13 dctx summed,
14 dS stored (arrival follows)."""
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
G, T, NSLOT = 256, tr.Td, 16
buf = torch.zeros(G * T * NSLOT, device="cuda", dtype=torch.int64)
n0 = lib.mmqg_decoder_persist_bwd_launch_count()
tr.g_dec.phase = 1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    check(lib.mmqg_decoder_persist_bwd_set_trace(buf.data_ptr(), buf.numel()))
    e0.record()
    check(lib.mmqg_decoder_seq_bwd(C.byref(tr.d_dec), C.byref(tr.g_dec), ops._stream()))
    e1.record()
    check(lib.mmqg_decoder_persist_bwd_set_trace(None, 0))
    torch.cuda.synchronize()
# the unstamped kernel, timed by HIP events (the loop + the two value-gradient kernels behind it)
torch.cuda.synchronize()
e0.record()
for rep in range(10):
    check(lib.mmqg_decoder_seq_bwd(C.byref(tr.d_dec), C.byref(tr.g_dec), ops._stream()))
e1.record()
torch.cuda.synchronize()
plain_us = e0.elapsed_time(e1) * 100.0
tr.g_dec.phase = 0
assert lib.mmqg_decoder_persist_bwd_launch_count() == n0 + 13, "the persistent decoder backward loop did not take this shape"
raw = buf.view(G, T, NSLOT).cpu()
assert int((raw[:, :, 0] != 0).sum()) == G * T, "the stamped instantiation did not run (no room for the stamps in LDS?)"
direct_mode = int((raw[:, :T - 1, 1] != 0).sum()) == 0        # no SW stage stamps
raw = (raw - raw[:, T - 1:T, 15:16].min()) % (1 << 32)      # low 32 bits of the 100 MHz counter, relative to the first start
t = raw.double() * 0.01                                   # us; token index = t (the loop runs T-1 .. 0)
t = t.flip(1)                                             # in execution order
print(f"decoder backward time loop, {T} tokens, {G} workgroups; all times in us; phase-1 call (loop + value gradients), unstamped kernel, "
      f"HIP events over 10 calls: {plain_us:.1f} per call")
per = t[:, 1:, 0].min(0).values - t[:, :-1, 0].min(0).values
print(f"whole loop (first start -> last dS stored): {float(t[:, -1, 14].max() - t[:, 0, 0].min()):.1f}")
print(f"prologue (kernel entry -> weights in LDS, barrier initialised, first token starts): mean {float((t[:, 0, 0] - t[:, 0, 15]).mean()):.1f}, "
      f"slowest workgroup {float((t[:, 0, 0] - t[:, 0, 15]).max()):.1f};  epilogue (last dS stored -> initial-state gradients stored): "
      f"{float(t[:, -1, 15].max() - t[:, -1, 14].max()):.1f};  kernel entry -> end: {float(t[:, -1, 15].max() - t[:, 0, 15].min()):.1f}")
print(f"token period: mean {float(per.mean()):.2f}  (tokens 2..{T - 2}: {float(per[2:-2].mean()):.2f})")
s = slice(2, T - 2)
names = [("wait for the attention barrier of the previous token", 0, 1), ("SW: dS W_attn_h partial tiles stored", 1, 2),
         ("barrier after SW (+ early loads of P2)", 2, 3), ("P2: late partials summed, cell backward", 3, 4), ("P2: late product stored", 4, 5),
         ("barrier after P2 (window: ahead product)", 5, 6), ("P1: late partials summed, cell backward", 6, 7), ("P1: late product stored", 7, 8),
         ("barrier after P1 (window: ahead product)", 8, 9), ("P0: late partials summed, cell backward", 9, 10), ("P0: late product (dctx) stored", 10, 11),
         ("barrier after P0 (window: ahead product, reduce)", 11, 12), ("ATT: dctx summed over the slices", 12, 13), ("ATT: value rows streamed, dS stored", 13, 14)]
if direct_mode:
    # the score-gradient product ran inside P2 (short score rows): no SW stage, the attention barrier is waited for by P2
    print("direct score-gradient product: no SW stage")
    names = [("barrier after the previous token's attention stage (+ early loads of P2)", 0, 3)] + names[3:]
for name, i0, i1 in names:
    d = t[:, s, i1] - t[:, s, i0]
    print(f"{name:56s} mean {float(d.mean()):6.2f}   slowest workgroup per token {float(d.max(0).values.mean()):6.2f}   fastest {float(d.min(0).values.mean()):6.2f}")
win = t[:, s, 14].max(0).values - t[:, s, 12].min(0).values
vals = 4 * tr.B * (tr.Lt * tr.H + tr.Lav * tr.Da + tr.Lav * tr.Dv)
print(f"attention-backward window (first workgroup past P0's barrier -> last dS stored): mean {float(win.mean()):.2f} us = "
      f"{vals / float(win.mean()) / 1e3:.0f} GB/s for the {vals / 1e6:.1f} MB of a token's value rows ({vals / float(win.mean()) / 8e6:.3f} of 8 TB/s)")
