#!/usr/bin/env python3
"""PCIe-inclusive training rate: the same steps as bench.py, but every step's batch starts in HOST memory (what the
reference's DataLoader hands over, train.py:149-162) and is copied to the device inside the timed region.

    python tools/host_batch_rate.py [workload] [steps]

bench.py's `value` is measured with the batch resident in HBM; this is the figure beside it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer

name = sys.argv[1] if len(sys.argv) > 1 else "config2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = WORKLOADS[name]
vid, text, dec = build_models(w, "cuda", seed=0)
tr = BatchedTrainer(vid, text, dec, batch_size=w.batch, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len,
                    use_graph=True).train()
host = synthetic_batch(w, seed=0)
nbytes = sum(v.numel() * v.element_size() for v in host.values())
variants = (("device-resident (bench.py)", {k: v.cuda() for k, v in host.items()}),
            ("host, pageable", host),
            ("host, pinned", {k: v.pin_memory() for k, v in host.items()}))
for label, batch in variants:
    for _ in range(5):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name} {label:28s} {dt * 1e3:7.3f} ms/step  {w.batch / dt:9.1f} questions/s   (batch {nbytes / 1e6:.2f} MB)")
