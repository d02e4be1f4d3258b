"""world_size-2 data-parallel tests on CPU (gloo): batch sharding, the bucketed gradient
all-reduce, and the DP identity 'two ranks on half batches + all-reduce == one rank on the
whole batch', with the oracle standing in for the compute (test infrastructure only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    import sys
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _worker_reducer(rank, world, port, q):
    _init(rank, world, port)
    import mmqg_amd  # noqa: F401
    from mmqg_amd.distributed import GradReducer, broadcast_parameters, shard_batch, trainer_buckets
    n = 1003
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    segs = {"dec": (0, 400), "text": (400, 700), "vid": (700, 800), "emb": (800, n)}
    red = GradReducer(flat, trainer_buckets(segs, n))
    red.reduce("dec")
    mid = flat.clone()                      # 'rest' not reduced yet: still this rank's values
    red.reduce("rest")
    red.finish()
    want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(flat, want) and torch.equal(mid[400:], torch.arange(n, dtype=torch.float32)[400:] * (rank + 1))
    # trainer layout dec | vid | text | emb (dec ends 2 floats short of the 4-aligned vid start): three buckets
    flat3 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    segs3 = {"dec": (0, 398), "vid": (400, 500), "text": (500, 800), "emb": (800, n)}
    b3 = trainer_buckets(segs3, n)
    ok = ok and b3 == [("dec", 0, 400), ("vid", 400, 500), ("rest", 500, n)]
    red3 = GradReducer(flat3, b3)
    red3.reduce("vid")
    red3.reduce("vid")                      # a second request for the same bucket is ignored
    red3.reduce_remaining()
    red3.finish()
    ok = ok and torch.equal(flat3, want)
    p = torch.full((10,), float(rank))
    broadcast_parameters(p)
    ok = ok and bool((p == 0).all()) and abs(red.grad_scale - 1.0 / world) < 1e-12
    batch = {"x": torch.arange(8).view(8, 1), "y": torch.arange(8)}
    sh = shard_batch(batch, rank, world)
    ok = ok and sh["y"].tolist() == list(range(rank * 4, rank * 4 + 4))
    q.put((rank, ok))
    dist.destroy_process_group()


def _worker_dp_identity(rank, world, port, q):
    _init(rank, world, port)
    import mmqg_amd  # noqa: F401
    from mmqg_amd.distributed import GradReducer, shard_batch
    from mmqg_amd.synthetic import Workload, build_models, synthetic_batch
    from oracle import mmqg_oracle as O
    w = Workload("dp", batch=4, n_frames=3, frame_dim=16, audio_dim=6, ctx_len=5, tgt_len=4, vocab=40, emb_dim=8,
                 hidden=12, layers=2, video_hidden=12, text_max_length=7, av_max_length=4, dropout=0.0)
    vid, text, dec = build_models(w, "cpu", seed=0)          # same seed: identical replicas
    cfg = dict(num_layers=w.layers, hidden_dim=w.hidden, text_max_length=w.text_max_length,
               av_max_length=w.av_max_length, video_hidden_dim=w.video_hidden, start_id=1, end_id=2, mask_mode=0)
    full = {k: (v.long() if v.dtype == torch.int32 else v) for k, v in synthetic_batch(w, seed=1, ragged=True).items()}

    def grads(batch):
        sd = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (dec, text, vid)]
        sd[1]["word_embeddings.weight"] = sd[0]["emb_layer.weight"]
        leaves = []
        for d in sd:
            for k, t in d.items():
                if t.is_floating_point() and "running" not in k and not any(t is x for x in leaves):
                    t.requires_grad_(True)
                    leaves.append(t)
        loss, _, _, _ = O.forward_loss(sd[0], sd[1], sd[2], batch, cfg, training=True)
        loss.backward()
        return torch.cat([(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1) for t in leaves]), float(loss)

    g_local, _ = grads(shard_batch(full, rank, world))
    red = GradReducer(g_local, [("dec", 0, g_local.numel() // 2), ("rest", g_local.numel() // 2, g_local.numel())])
    red.reduce("dec"); red.reduce("rest"); red.finish()
    g_dp = g_local * red.grad_scale
    g_full, _ = grads(full)
    err = float((g_dp - g_full).abs().max())
    q.put((rank, err <= 1e-5 * max(1.0, float(g_full.abs().max())), err))
    dist.destroy_process_group()


def _run(worker, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_bucketed_allreduce_broadcast_and_sharding_world2():
    for r in _run(_worker_reducer):
        assert r[1], f"rank {r[0]} failed"


def test_two_ranks_on_half_batches_equal_one_rank_on_the_full_batch():
    for r in _run(_worker_dp_identity):
        assert r[1], f"rank {r[0]}: max gradient difference {r[2]}"


def test_shard_batch_rejects_indivisible_batches():
    import mmqg_amd  # noqa: F401
    from mmqg_amd.distributed import shard_batch
    with pytest.raises(ValueError):
        shard_batch({"x": torch.zeros(5, 2)}, 0, 2)
