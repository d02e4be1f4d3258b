"""Helpers that turn the committed .npz fixtures into oracle-style parameter dicts and batches."""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import mmqg_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Parity bar shared by every test file: max |got - want| <= tol * max|want|, i.e. RELATIVE to the expected
# tensor's own magnitude (attention weights ~1/283 and small gradients are held to 1e-4 of themselves, not
# of 1.0), with an absolute floor for tensors that are (near) zero.  Every comparison is logged so the
# observed worst errors can be read back (gpurun_out/parity_errors.tsv on the GPU box).
ABS_FLOOR = 1e-7
PARITY_LOG = []


def close(got, want, tol=1e-4, what="", floor=ABS_FLOOR):
    got = torch.as_tensor(np.asarray(got) if not torch.is_tensor(got) else got).detach().double().cpu()
    want = torch.as_tensor(np.asarray(want) if not torch.is_tensor(want) else want).detach().double().cpu()
    assert got.shape == want.shape, (what, tuple(got.shape), tuple(want.shape))
    if want.numel() == 0:
        return
    assert bool(torch.isfinite(got).all()), f"{what}: non-finite values"
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    allowed = max(tol * scale, floor)
    PARITY_LOG.append((what, err, scale, err / scale if scale > 0 else 0.0, tol))
    assert err <= allowed, f"{what}: max abs err {err:.3e} > {allowed:.3e} (tol {tol:g} x max|want| {scale:.3e})"


def dump_parity_log(path):
    if not PARITY_LOG:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a") as f:
        for what, err, scale, rel, tol in PARITY_LOG:
            f.write(f"{what}\t{err:.3e}\t{scale:.3e}\t{rel:.3e}\t{tol:g}\n")
    PARITY_LOG.clear()


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def state_from(z, prefix):
    """All arrays under ``prefix/`` as an ordered dict of tensors (state-dict key names)."""
    out = OrderedDict()
    plen = len(prefix) + 1
    for k in z.files:
        if k.startswith(prefix + "/"):
            out[k[plen:]] = torch.from_numpy(np.array(z[k]))
    return out


def small_cfg(z):
    c = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg/")}
    return c, dict(num_layers=c["L"], hidden_dim=c["H"], text_max_length=c["Lt"], av_max_length=c["Lav"],
                   video_hidden_dim=c["Dv"], start_id=c["start_id"], end_id=c["end_id"], mask_mode=0)


def small_params(z, prefix="init"):
    vid = state_from(z, f"{prefix}/vid")
    text = state_from(z, f"{prefix}/text")
    dec = state_from(z, f"{prefix}/dec")
    text["word_embeddings.weight"] = dec["emb_layer.weight"]          # one shared table
    return dec, text, vid


def small_samples(z, n=3):
    out = []
    for b in range(n):
        out.append({k: torch.from_numpy(np.array(z[f"in/{b}/{k}"])) for k in ("frames", "audio", "context", "target")})
    return out


def collate(samples, pad_id=0):
    """Pad a list of per-question dicts into one batch (frames go through the reference's view)."""
    B = len(samples)
    n_frames = torch.tensor([s["frames"].shape[1] if s["frames"].dim() == 4 else s["frames"].shape[0] for s in samples])
    ctx_len = torch.tensor([len(s["context"]) for s in samples])
    tgt_len = torch.tensor([len(s["target"]) for s in samples])
    T, Tc, Td = int(n_frames.max()), int(ctx_len.max()), int(tgt_len.max())
    if samples[0]["frames"].dim() == 4:
        frames = torch.stack([O.view_frames_like_reference(s["frames"], T) for s in samples])
    else:
        frames = torch.stack([torch.nn.functional.pad(s["frames"], (0, 0, 0, T - s["frames"].shape[0])) for s in samples])
    audio = torch.stack([torch.nn.functional.pad(s["audio"], (0, 0, 0, T - s["audio"].shape[0])) for s in samples])
    context = torch.full((B, Tc), pad_id, dtype=torch.long)
    target = torch.full((B, Td), pad_id, dtype=torch.long)
    for b, s in enumerate(samples):
        context[b, :ctx_len[b]] = s["context"]
        target[b, :tgt_len[b]] = s["target"]
    return dict(frames=frames, audio=audio, context=context, ctx_len=ctx_len, target=target, tgt_len=tgt_len,
                n_frames=n_frames)


def clone_params(*sds):
    """Deep copies that keep the embedding shared between the text and decoder dicts."""
    dec, text, vid = sds
    dec2 = OrderedDict((k, v.clone()) for k, v in dec.items())
    text2 = OrderedDict((k, v.clone()) for k, v in text.items())
    vid2 = OrderedDict((k, v.clone()) for k, v in vid.items())
    text2["word_embeddings.weight"] = dec2["emb_layer.weight"]
    return dec2, text2, vid2
