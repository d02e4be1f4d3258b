#!/usr/bin/env python3
"""One training step (zero_grad -> forward -> loss -> backward, dropout live) of a BASELINE workload with fixed
seeds; writes the loss, the whole flat gradient and the logits of a few questions to an .npz file.

    python tools/step_dump.py --workload config2 --batch 64 --out /tmp/a.npz [--graph]

The A/B switches of the library (MMQG_GEMM_X3, MMQG_NO_PERSIST, MMQG_NO_WIDE, ...) are read once per process, so a
test that compares two kernel families gradient by gradient runs this script once per setting
(tests/test_hip_model.py::test_kernel_families_give_the_same_gradients_at_bench_size)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--graph", action="store_true", help="take the step through the captured hipGraph (incl. Adam)")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--force-dp", action="store_true",
                    help="ONE rank through the data-parallel schedule (process group over RCCL, MMQG_FORCE_DP=1): the cut "
                         "graphs, the bucket all-reduces between them, early / late Adam")
    ap.add_argument("--questions", type=int, default=4, help="how many questions' logits to keep")
    a = ap.parse_args()
    import mmqg_amd  # noqa: F401
    from mmqg_amd import _lib
    from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
    from mmqg_amd.trainer import BatchedTrainer
    w = WORKLOADS[a.workload]
    B = a.batch or w.batch
    dev = torch.device("cuda", 0)
    if a.force_dp:
        import socket
        from mmqg_amd.distributed import configure_rccl_env
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(MMQG_FORCE_DP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        configure_rccl_env()
        torch.cuda.set_device(dev)
        torch.distributed.init_process_group("nccl", device_id=dev)
    vid, text, dec = build_models(w, dev, seed=3)
    tr = BatchedTrainer(vid, text, dec, batch_size=B, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len,
                        lr=1e-4, seed=4321, use_graph=a.graph).train()
    batch = {k: v.to(dev) for k, v in synthetic_batch(w, seed=17, batch=B, ragged=a.ragged).items()}
    assert tr.distributed == bool(a.force_dp)
    logits = tr.forward_only(batch, training=True)[:a.questions].cpu().numpy().copy()
    out = {"logits": logits}
    if a.graph or a.force_dp:
        p0 = tr.flat_p.clone()
        loss = float(tr.step(batch))        # (data parallel: the eager or the cut-graph schedule, incl. the exchange and Adam)
        torch.cuda.synchronize()
        tr.check_health(sync=True)
        out["dp"] = (tr.flat_p - p0).cpu().numpy()          # one Adam step of the whole model
    else:
        loss = float(tr.forward_backward(batch))
    out["loss"] = np.float64(loss)
    g = tr.flat_g.cpu().numpy()
    for name, (lo, hi) in tr.segments.items():
        out["grad_" + name] = g[lo:hi]
    out["persist_launches"] = np.int64(_lib.load().mmqg_persist_launch_count())
    out["persist_bwd_launches"] = np.int64(_lib.load().mmqg_persist_bwd_launch_count())
    out["decoder_persist_launches"] = np.int64(_lib.load().mmqg_decoder_persist_launch_count())
    out["decoder_persist_bwd_launches"] = np.int64(_lib.load().mmqg_decoder_persist_bwd_launch_count())
    out["projection_kernel"] = np.int64(_lib.load().mmqg_projection_last_kernel())
    out["persist_declined"] = np.int64(_lib.load().mmqg_persist_declined_count())
    out["persist_failures"] = np.int64(_lib.load().mmqg_persist_failures())
    np.savez(a.out, **out)
    if a.force_dp:
        torch.distributed.destroy_process_group()
    print(f"step_dump: {w.name.split(':')[0]} B={B} loss {loss:.6f} -> {a.out}")


if __name__ == "__main__":
    main()
