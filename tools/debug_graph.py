import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd
from mmqg_amd.synthetic import WORKLOADS, Workload, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer

def run(w, use_graph, steps, tag=""):
    vid, text, dec = build_models(w, "cuda", seed=0)
    tr = BatchedTrainer(vid, text, dec, batch_size=w.batch, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len, seed=1234, use_graph=use_graph).train()
    bs = [{k: v.cuda() for k, v in synthetic_batch(w, seed=i).items()} for i in range(4)]
    first_bad = None
    for i in range(steps):
        loss = float(tr.step(bs[i % 4]))
        if loss != loss:
            first_bad = i
            break
    print(tag, "graph" if use_graph else "eager", "steps", steps, "first NaN at", first_bad, "last loss", loss, flush=True)

mode = sys.argv[1]
steps = int(sys.argv[2])
run(WORKLOADS["config2"], mode == "graph", steps, tag=os.environ.get("TAG", ""))
