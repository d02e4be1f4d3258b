#!/usr/bin/env python3
"""Do two independent half-batch steps, replayed on two streams, finish sooner than one full-batch step?
(The time loops are chains of latency-bound launches: two chains could fill each other's gaps.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer

w = WORKLOADS["config2"]
dev = torch.device("cuda", 0)

def make(B, seed):
    vid, text, dec = build_models(w, dev, seed=seed)
    tr = BatchedTrainer(vid, text, dec, batch_size=B, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len,
                        lr=1e-4, seed=1, use_graph=True).train()
    b = {k: v.to(dev) for k, v in synthetic_batch(w, seed=seed, batch=B).items()}
    return tr, b

def run(trs, streams, steps=20):
    for _ in range(3):
        for (tr, b), s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (tr, b), s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

one = [make(64, 0)]
ms = run(one, [torch.cuda.Stream()])
print(f"one trainer,  B=64            : {ms:.3f} ms per 64 questions")
os.environ["MMQG_NO_PERSIST"] = "1"      # two persistent launches must not overlap
two = [make(32, 1), make(32, 2)]
ms1 = run(two[:1], [torch.cuda.Stream()])
print(f"one trainer,  B=32 (launches) : {ms1:.3f} ms per 32 questions")
ms2 = run(two, [torch.cuda.Stream(), torch.cuda.Stream()])
print(f"two trainers, B=32 each, two streams: {ms2:.3f} ms per 64 questions")
