#!/usr/bin/env python3
"""PCIe-inclusive training rate: the same steps as bench.py, but every step's batch starts in HOST memory (what the
reference's DataLoader hands over, train.py:149-162) and is copied to the device inside the timed region.

    python tools/host_batch_rate.py [workload] [steps] [modes: dma,mapped,copy]

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
# what the CPU side of a staged batch costs: torch's (multi-threaded) copy_ against one memcpy, into pinned memory
import ctypes
src = host["frames"].contiguous()
dst = torch.empty_like(src).pin_memory()
for label, fn in (("torch copy_ into pinned", lambda: dst.copy_(src)),
                  ("memmove into pinned", lambda: ctypes.memmove(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size()))):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    print(f"{name} CPU staging copy of the frames ({src.numel() * 4 / 1e6:.2f} MB), {label:26s}: {(time.perf_counter() - t0) * 100:.3f} ms "
          f"(torch threads {torch.get_num_threads()}, cpus visible {os.cpu_count()}, usable {len(os.sched_getaffinity(0))})", flush=True)
hosts = [synthetic_batch(w, seed=i) for i in range(4)]          # four different batches, as a DataLoader would hand over
variants = [("device-resident (bench.py)", None, [{k: v.cuda() for k, v in h.items()} for h in hosts])]
for mode in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("dma", "mapped", "copy")):
    variants.append((f"host, pageable [{mode}]", mode, hosts))
    variants.append((f"host, pinned   [{mode}]", mode, [{k: v.pin_memory() for k, v in h.items()} for h in hosts]))
for label, mode, batches in variants:
    if mode:
        os.environ["MMQG_HOST_BATCH"] = mode
    for i in range(5):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name} {label:36s} {dt * 1e3:7.3f} ms/step  {w.batch / dt:9.1f} questions/s   (batch {nbytes / 1e6:.2f} MB)", flush=True)
