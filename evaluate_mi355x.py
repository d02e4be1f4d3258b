#!/usr/bin/env python3
"""Batched MI355X counterpart of the reference's ``evaluate.py`` (evaluate() :34-127, __main__ :129-219):
same command line, same checkpoint files in, same predictions file out.

    python evaluate_mi355x.py -c results/test/config.json -s greedy -b      # best-epoch checkpoints
    python evaluate_mi355x.py -c results/test/config.json -s sampling -l    # last-epoch checkpoints

The free-running decode (greedy / sampling / topk — the reference's top-k is k=1, evaluate.py:96) runs
for a whole batch of questions on the device (``BatchedTrainer.decode`` -> ``mmqg_decoder_decode_run``);
predictions are cut at the first ``<end>`` (evaluate.py:101-103) and scored with the reference's BLEU
call (evaluate.py:108-112).  Output: ``{best|last}_predictions_{strategy}.json`` with the reference's
records {question_id, gt_question, pred_question}.  Like evaluate.py:163 (and unlike train.py:229) the
frames are normalised with config.vid_mean / vid_std; ``--no-normalize`` turns that off.
"""
from __future__ import annotations

import argparse
import json

import numpy as np
import torch

import mmqg_amd  # noqa: F401
from mmqg_amd.checkpoint import load_models
from mmqg_amd.config import Config
from mmqg_amd.metrics import ids_to_words, reference_bleu_scores, truncate_at_end
from mmqg_amd.trainer import BatchedTrainer
from train_mi355x import build_from_config


def evaluate(trainer, dataset, batch_size, Tf, Tc, Td, audio_dim, strategy, seed=0):
    """-> (predictions, bleu, bleu_1, bleu_2, bleu_3) as evaluate.py:34-127 returns them."""
    from mmqg_amd.data import collate_questions
    end_id = dataset.vocab["<end>"]
    predictions, tot = [], {"bleu": 0.0, "bleu_1": 0.0, "bleu_2": 0.0, "bleu_3": 0.0}
    n = len(dataset)
    for i in range(0, n, batch_size):
        idx = list(range(i, min(i + batch_size, n)))
        pad = idx + [idx[-1]] * (batch_size - len(idx))             # last batch: repeat, then ignore the copies
        batch = collate_questions([dataset[j] for j in pad], [dataset.audio_features(j) for j in pad], Tf, Tc, Td, audio_dim)
        out = trainer.decode({k: v for k, v in batch.items() if torch.is_tensor(v)}, max_len=Td,
                             strategy="greedy" if strategy == "topk" else strategy, seed=seed + i)
        ids = out["ids"].cpu()
        for b in range(len(idx)):
            words = ids_to_words(truncate_at_end(ids[b].tolist(), end_id), dataset.index_to_word)
            for k, v in reference_bleu_scores(batch["question"][b], words).items():
                tot[k] += v
            qid = batch["question_id"][b]
            predictions.append({"question_id": int(qid) if str(qid).lstrip("-").isdigit() else qid,
                                "gt_question": batch["question"][b], "pred_question": " ".join(words)})
    return (predictions,) + tuple(tot[k] / max(n, 1) for k in ("bleu", "bleu_1", "bleu_2", "bleu_3"))


def main(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate model")
    ap.add_argument("-b", "--best", action="store_true", help="get best epoch results")
    ap.add_argument("-l", "--last", action="store_true", help="get last epoch results")
    ap.add_argument("-c", "--config_path", type=str, required=True)
    ap.add_argument("-s", "--strategy", type=str, required=True, choices=["greedy", "sampling", "topk"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--max-frames", type=int, default=16)
    ap.add_argument("--max-context", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-normalize", action="store_true")
    a = ap.parse_args(argv)
    from mmqg_amd.data import Compose, Normalize, Resize, ToFloatTensor, VQGDataset, prepare_sequence
    cfg = Config(a.config_path)
    dev = torch.device("cuda", 0)
    tfm = [ToFloatTensor(), Resize(112)]
    if not a.no_normalize:
        tfm.append(Normalize(cfg.vid_mean, cfg.vid_std))
    ds = VQGDataset(cfg.test_file, cfg.vocab_file, cfg.index_to_word_file, str(cfg.salient_frames_path),
                    str(cfg.salient_audio_path), prepare_sequence, Compose(tfm))
    w_path = cfg.output_path / "last_weigths.pt" if a.last else cfg.learned_weight_path        # evaluate.py:168-171
    weights = torch.load(w_path, map_location="cpu", weights_only=False)
    weights = weights.detach().float() if torch.is_tensor(weights) else torch.as_tensor(np.asarray(weights)).float()
    av, text, dec = build_from_config(cfg, weights, dev)
    load_models(cfg, av, text, dec, last=a.last)
    B = a.batch or cfg.batch_size
    Td = cfg.question_max_length
    tr = BatchedTrainer(av, text, dec, batch_size=B, n_frames=a.max_frames, ctx_len=a.max_context, tgt_len=Td,
                        start_id=ds.vocab["<start>"], seed=a.seed).eval()
    preds, bleu, b1, b2, b3 = evaluate(tr, ds, B, a.max_frames, a.max_context, Td, cfg.audio_emb, a.strategy, a.seed)
    print(f"Val_bleu - {round(bleu, 3)}, Val_bleu_1 - {round(b1, 3)}")
    out_file = cfg.output_path / f"{'last' if a.last else 'best'}_predictions_{a.strategy}.json"
    with open(out_file, "w") as f:
        json.dump(preds, f)
    print(f"Predictions saved to {out_file}")
    return preds, bleu, b1, b2, b3


if __name__ == "__main__":
    main()
