#!/usr/bin/env python3
"""Batched MI355X training driver: the counterpart of the reference's ``train.py`` (__main__
train.py:220-293, train() :131-218, validate() :61-129) on the HIP path.

    python train_mi355x.py --synthetic config2 --steps 50            # synthetic tensors, questions/s
    python train_mi355x.py --config results/test/config.json         # real data in the reference's formats
    python -m torch.distributed.run --nproc-per-node 8 train_mi355x.py --synthetic config2

Real-data mode reads the files the reference's prep scripts write (Config paths): split JSONs,
vocab / index_to_word JSON, weight_matrix.npy (GloVe), ``v_{vid}_q_{qid}_.npy`` frames and — because
the VGGish front-end is outside this build — ``v_{vid}_q_{qid}_.npy`` audio features next to the wavs.
Per epoch: batched steps, then validate() semantics (free-running greedy decode, per-step CE, the
reference's BLEU variant), best / last checkpoints under the reference's file names.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch

import mmqg_amd  # noqa: F401
from mmqg_amd.checkpoint import save_models, save_training_state
from mmqg_amd.config import Config
from mmqg_amd.metrics import ids_to_words, reference_bleu_scores, truncate_at_end
from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer


def setup_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from mmqg_amd.distributed import configure_rccl_env
        configure_rccl_env()           # RCCL's channels fit the CUs the persistent backward loop leaves free
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    return world, rank, torch.device("cuda", local)


def run_synthetic(a, world, rank, dev):
    w = WORKLOADS[a.synthetic]
    B = a.batch or w.batch
    vid, text, dec = build_models(w, dev, seed=a.seed)
    tr = BatchedTrainer(vid, text, dec, batch_size=B, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len,
                        lr=a.lr, seed=a.seed, use_graph=not a.no_graph).train()
    batches = [{k: v.to(dev) for k, v in synthetic_batch(w, seed=rank * 1000 + i, batch=B).items()} for i in range(8)]
    t0 = None
    for i in range(a.steps):
        if i == min(5, a.steps - 1):
            torch.cuda.synchronize(); t0, i0 = time.perf_counter(), i
        loss = tr.step(batches[i % len(batches)])
        if rank == 0 and (i % a.log_every == 0 or i == a.steps - 1):
            print(f"step {i:5d}  loss/token {tr.loss_value() / w.tgt_len:.4f}", flush=True)
    torch.cuda.synchronize()
    if rank == 0 and t0 is not None and a.steps - i0 > 0:
        dt = time.perf_counter() - t0
        print(f"{world * B * (a.steps - i0) / dt:.1f} questions/s over {a.steps - i0} steps on {world} GPU(s)")


def build_from_config(cfg, weights, dev):
    """The three modules as train.py:236-258 / evaluate.py:175-207 construct them, sharing one embedding."""
    import importlib
    enc = importlib.import_module("multi-modal-qg_amd.model.encoder")
    decm = importlib.import_module("multi-modal-qg_amd.model.decoder")
    emb = torch.nn.Embedding(*weights.shape)
    emb.load_state_dict({"weight": weights})
    av = enc.AudioVideoEncoder(cfg.av_in_channels, cfg.av_kernel_sz, cfg.av_stride, cfg.video_hidden_dim, cfg.flatten_dim)
    text = enc.TextEncoder(cfg.text_lstm_layers, cfg.text_lstm_dropout, cfg.text_lstm_hidden_dim, weights.shape[1], emb, dev)
    dec = decm.AttnDecoder(cfg.dec_lstm_layers, cfg.dec_lstm_dropout, cfg.dec_lstm_hidden_dim, weights.shape[0],
                           weights.shape[1], cfg.video_hidden_dim, cfg.audio_emb, emb, cfg.context_max_lenth,
                           cfg.av_max_length, dev)
    dec.mask_mode = cfg.attention_mask_mode
    for m in (av, text, dec):
        m.to(dev)
    return av, text, dec


def run_real(a, world, rank, dev):
    from mmqg_amd.data import Compose, Resize, ToFloatTensor, VQGDataset, collate_questions, prepare_sequence
    cfg = Config(a.config)
    weights = torch.from_numpy(np.load(cfg.weights_matrix_file)).float()      # kept float (train.py:227 truncates to int)
    tfm = Compose([ToFloatTensor(), Resize(112)])
    train_ds = VQGDataset(cfg.train_file, cfg.vocab_file, cfg.index_to_word_file, str(cfg.salient_frames_path),
                          str(cfg.salient_audio_path), prepare_sequence, tfm)
    val_ds = VQGDataset(cfg.val_file, cfg.vocab_file, cfg.index_to_word_file, str(cfg.salient_frames_path),
                        str(cfg.salient_audio_path), prepare_sequence, tfm)
    av, text, dec = build_from_config(cfg, weights, dev)
    B = a.batch or cfg.batch_size
    Tf, Tc, Td = a.max_frames, a.max_context, cfg.question_max_length
    tr = BatchedTrainer(av, text, dec, batch_size=B, n_frames=Tf, ctx_len=Tc, tgt_len=Td, lr=cfg.lr,
                        start_id=train_ds.vocab["<start>"], seed=a.seed)
    end_id = train_ds.vocab["<end>"]

    def train_batches(epoch):
        order = np.random.default_rng(a.seed + epoch).permutation(len(train_ds))
        gb = B * world
        for i in range(0, len(order) - gb + 1, gb):                               # drop the ragged tail
            idx = order[i:i + gb][rank * B:(rank + 1) * B]
            yield collate_questions([train_ds[j] for j in idx], [train_ds.audio_features(j) for j in idx], Tf, Tc, Td,
                                    cfg.audio_emb)

    def val_batches():
        """Every validation question exactly once over all ranks: rank r takes questions r, r+world, ...;
        a short last batch is padded by repeating its last question and the copies are ignored."""
        mine = list(range(rank, len(val_ds), world))
        for i in range(0, len(mine), B):
            idx = mine[i:i + B]
            pad = idx + [idx[-1]] * (B - len(idx))
            yield len(idx), collate_questions([val_ds[j] for j in pad], [val_ds.audio_features(j) for j in pad], Tf, Tc,
                                              Td, cfg.audio_emb)

    def all_sum(vals):
        t = torch.tensor(vals, device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t)
        return t.tolist()

    def sync_bn_buffers():
        """BatchNorm running statistics advance on every rank's own shard; checkpoints hold their mean."""
        if world > 1:
            for name, buf in av.named_buffers():
                if buf.is_floating_point():
                    torch.distributed.all_reduce(buf)
                    buf.div_(world)

    best = float("inf")
    bleu_keys = ("bleu", "bleu_1", "bleu_2", "bleu_3")
    for epoch in range(cfg.epochs if a.epochs is None else a.epochs):
        tr.train()
        tot, n = 0.0, 0
        for batch in train_batches(epoch):
            loss = tr.step({k: v for k, v in batch.items() if torch.is_tensor(v)})
            tot += tr.loss_value() / max(1.0, float(batch["tgt_len"].float().mean())); n += 1
        tr.eval()
        sync_bn_buffers()
        # validate() (train.py:61-129): per question loss / target_len and the four BLEU numbers, averaged over
        # the questions of the whole validation set (all ranks)
        vloss, bleu, m = 0.0, dict.fromkeys(bleu_keys, 0.0), 0
        for valid, batch in val_batches():
            out = tr.decode({k: v for k, v in batch.items() if torch.is_tensor(v)}, with_loss=True)
            per_q = (out["loss_per_question"].cpu() / batch["tgt_len"].float().clamp(min=1))[:valid]
            vloss += float(per_q.sum()); m += valid
            for b in range(valid):
                pred = ids_to_words(truncate_at_end(out["ids"][b].tolist(), end_id), val_ds.index_to_word)
                for k, v in reference_bleu_scores(batch["question"][b], pred).items():
                    bleu[k] += v
        tr.check_health(sync=True)            # never checkpoint behind a failed persistent launch
        tot, n, vloss, m, *bl = all_sum([tot, n, vloss, m] + [bleu[k] for k in bleu_keys])
        val_loss = vloss / max(m, 1.0)
        if rank == 0:
            stats = {"epoch": epoch, "train_loss": tot / max(n, 1.0), "val_loss": val_loss,
                     **{k: v / max(m, 1.0) for k, v in zip(bleu_keys, bl)}}
            print(json.dumps(stats), flush=True)
            if m and val_loss < best:                                             # train.py:198-206
                best = val_loss
                save_models(cfg, av, text, dec)
            save_models(cfg, av, text, dec, last=True)                            # train.py:209-214
            save_training_state(cfg.output_path / "training_state.pt", tr, epoch=epoch)
    if rank == 0:
        cfg.save_config()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", default=None, help="workload name (config1, config2, config2-tight, config4, config5)")
    ap.add_argument("--config", default=None, help="config.json in the reference's format (real data)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-frames", type=int, default=16)
    ap.add_argument("--max-context", type=int, default=128)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    world, rank, dev = setup_distributed()
    if a.config:
        run_real(a, world, rank, dev)
    else:
        a.synthetic = a.synthetic or "config2"
        run_synthetic(a, world, rank, dev)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
