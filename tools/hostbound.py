import sys, time, torch
sys.path.insert(0, '/root/repo')
import mmqg_amd
from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer
w = WORKLOADS["config2"]
vid, text, dec = build_models(w, "cuda", seed=0)
tr = BatchedTrainer(vid, text, dec, batch_size=w.batch, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len, use_graph=False).train()
b = {k: v.cuda() for k, v in synthetic_batch(w, seed=0).items()}
for _ in range(5): tr.step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): tr.step(b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host enqueue {1e3*(t1-t0)/40:.2f} ms/step, total {1e3*(t2-t0)/40:.2f} ms/step")
