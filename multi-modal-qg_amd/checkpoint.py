"""Checkpoint files with the reference's names and contents (train.py:198-214, config.py:19-25;
reloaded by evaluate.py:168-211), plus what the reference lacks: optimizer state and resume.

best epoch : av_model.pth, text_enc_model.pth, dec_model.pth (state_dicts), learned_weight.pt (the
             shared embedding Parameter)
last epoch : last_av_model.pth, last_text_enc.pth, last_decoder.pth, last_weigths.pt [sic]
"""
from __future__ import annotations

import os
from pathlib import Path

import torch


def _paths(config, last: bool):
    out = Path(config.output_path)
    if last:
        return out / "last_av_model.pth", out / "last_text_enc.pth", out / "last_decoder.pth", out / "last_weigths.pt"
    return Path(config.av_model_path), Path(config.text_enc_model_path), Path(config.dec_model_path), Path(config.learned_weight_path)


def _cpu(sd):
    return {k: v.detach().cpu().clone() for k, v in sd.items()}


def save_models(config, av_enc_model, text_enc_model, dec_model, last: bool = False) -> None:
    av_p, text_p, dec_p, w_p = _paths(config, last)
    os.makedirs(av_p.parent, exist_ok=True)
    torch.save(_cpu(av_enc_model.state_dict()), av_p)
    torch.save(_cpu(text_enc_model.state_dict()), text_p)
    torch.save(_cpu(dec_model.state_dict()), dec_p)
    torch.save(torch.nn.Parameter(dec_model.emb_layer.weight.detach().cpu().clone()), w_p)


def load_models(config, av_enc_model, text_enc_model, dec_model, last: bool = False, map_location="cpu") -> None:
    """Loads in place (parameters that are views of a trainer's flat buffer stay views).  Keys of
    the reference's VGGish sub-module (``audio_enc.vggish.*``) are ignored: that front-end is not
    part of this build."""
    av_p, text_p, dec_p, _ = _paths(config, last)
    av_sd = torch.load(av_p, map_location=map_location)
    av_sd = {k: v for k, v in av_sd.items() if not k.startswith("audio_enc.")}
    av_enc_model.load_state_dict(av_sd, strict=False)
    text_enc_model.load_state_dict(torch.load(text_p, map_location=map_location))
    dec_model.load_state_dict(torch.load(dec_p, map_location=map_location))


def save_training_state(path, trainer, epoch: int = 0, extra: dict = None) -> None:
    """Everything needed to resume bit-for-bit: flat parameters, both Adam moment sets, the step
    counter that also drives the dropout streams, BatchNorm running statistics."""
    bufs = {k: v.detach().cpu().clone() for k, v in trainer.video.state_dict().items() if "running" in k or "num_batches" in k}
    torch.save({"flat_p": trainer.flat_p.cpu(), "flat_m": trainer.flat_m.cpu(), "flat_v": trainer.flat_v.cpu(),
                "emb_m2": trainer.emb_m2.cpu(), "emb_v2": trainer.emb_v2.cpu(), "step": int(trainer.step_dev.item()),
                "seed": trainer.seed, "segments": trainer.segments, "epoch": epoch, "bn_buffers": bufs,
                "extra": extra or {}}, path)


def load_training_state(path, trainer) -> dict:
    st = torch.load(path, map_location="cpu", weights_only=False)
    if st["segments"] != trainer.segments:
        raise ValueError("checkpoint was written for a different parameter layout")
    for name in ("flat_p", "flat_m", "flat_v", "emb_m2", "emb_v2"):
        getattr(trainer, name).copy_(st[name])
    trainer.counters.fill_(st["step"])       # completed steps = Adam's step number
    trainer.set_seed(st["seed"])
    trainer.video.load_state_dict(st["bn_buffers"], strict=False)
    return st
