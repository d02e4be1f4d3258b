#!/usr/bin/env python3
"""Free-running greedy decode (validate()/evaluate() path, train.py:100-110) at a workload's shapes:
questions/s for encoders + max_len decoder steps, all on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "config2"]
vid, text, dec = build_models(w, "cuda", seed=0)
tr = BatchedTrainer(vid, text, dec, batch_size=w.batch, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len).eval()
b = {k: v.cuda() for k, v in synthetic_batch(w, seed=0).items()}
T = 21
for strategy in ("greedy", "sampling"):
    for _ in range(3):
        tr.decode(b, max_len=T, strategy=strategy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        tr.decode(b, max_len=T, strategy=strategy)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{w.name.split(':')[0]} {strategy:8s}: {dt * 1e3:.2f} ms per batch of {w.batch} x {T} tokens = {w.batch / dt:,.0f} questions/s")
