"""Synthetic workloads for benchmarks and parity tests: seeded frame / audio /
text tensors of the shapes BASELINE.json's configs name, and random-initialised modules of the
reference architecture (reference init: orthogonal LSTM matrices, N(0,1) biases,
Xavier-uniform Linear weights — decoder.py:109-123, encoder.py:73-78,102-107; embedding table
N(0, 0.6^2) as prepare_data.py:42 draws unknown words)."""
from __future__ import annotations

from dataclasses import dataclass, asdict

import torch
from torch import nn


@dataclass
class Workload:
    name: str
    batch: int = 64              # questions per GPU
    n_frames: int = 8
    frame_dim: int = 2048        # per-frame feature width (0 -> raw 3x112x112 frames through the CNN)
    audio_dim: int = 128
    ctx_len: int = 32
    tgt_len: int = 20
    vocab: int = 10000
    emb_dim: int = 300
    hidden: int = 512
    layers: int = 3
    video_hidden: int = 512
    text_max_length: int = 283   # config.py:70 context_max_lenth
    av_max_length: int = 101     # config.py:71
    dropout: float = 0.2
    image: int = 112

    def dict(self):
        return asdict(self)


# BASELINE.json configs.  "padded" = attention widths of config.py (what train.py allocates);
# "tight" = widths equal to the actual sequence lengths.
WORKLOADS = {
    "config1": Workload("config1: config.py defaults, batch 4, raw frames", batch=4, frame_dim=0),
    "config2": Workload("config2: B=64, 8 frames x2048, 8 audio x128, 32 ctx tokens, 20-token decode, V=10k"),
    "config2-tight": Workload("config2 with attention widths 32/8", text_max_length=32, av_max_length=8),
    "config4": Workload("config4: long context, 32 frames + 128 ctx tokens, 40-token decode, B=32", batch=32,
                        n_frames=32, ctx_len=128, tgt_len=40),
    "config5": Workload("config5: large vocab, V=50k, H=1024, B=128", batch=128, vocab=50000, hidden=1024,
                        video_hidden=1024),
}


def build_models(w: Workload, device, seed: int = 0):
    """(frame encoder, text encoder, decoder) of the drop-in classes, on ``device``."""
    import importlib
    enc = importlib.import_module(__package__ + ".model.encoder")
    dec_m = importlib.import_module(__package__ + ".model.decoder")
    torch.manual_seed(seed)
    emb = nn.Embedding(w.vocab, w.emb_dim)
    with torch.no_grad():
        emb.weight.normal_(0.0, 0.6)
    feat = w.frame_dim if w.frame_dim else 10 * (((w.image - 4) // 3 - 4) // 3) ** 2
    vid = enc.VideoConvLstmEncoder(3, 3, 1, w.video_hidden, feat)
    text = enc.TextEncoder(w.layers, w.dropout, w.hidden, w.emb_dim, emb, device)
    dec = dec_m.AttnDecoder(w.layers, w.dropout, w.hidden, w.vocab, w.emb_dim, w.video_hidden, w.audio_dim, emb,
                            w.text_max_length, w.av_max_length, device)
    return vid.to(device), text.to(device), dec.to(device)


def synthetic_batch(w: Workload, seed: int = 0, batch: int = None, ragged: bool = False) -> dict:
    """CPU tensors; ids 0/1/2 are reserved (<pad>,<start>,<end>, prepare_data.py:63-66)."""
    g = torch.Generator().manual_seed(seed)
    B = batch or w.batch
    if w.frame_dim:
        frames = torch.randn(B, w.n_frames, w.frame_dim, generator=g)
    else:
        frames = torch.rand(B, w.n_frames, 3, w.image, w.image, generator=g)
    audio = torch.randn(B, w.n_frames, w.audio_dim, generator=g)
    context = torch.randint(3, w.vocab, (B, w.ctx_len), generator=g)
    target = torch.randint(3, w.vocab, (B, w.tgt_len), generator=g)
    if ragged:
        n_frames = torch.randint(max(1, w.n_frames // 2), w.n_frames + 1, (B,), generator=g)
        ctx_len = torch.randint(max(1, w.ctx_len // 2), w.ctx_len + 1, (B,), generator=g)
        tgt_len = torch.randint(max(1, w.tgt_len // 2), w.tgt_len + 1, (B,), generator=g)
    else:
        n_frames = torch.full((B,), w.n_frames)
        ctx_len = torch.full((B,), w.ctx_len)
        tgt_len = torch.full((B,), w.tgt_len)
    for b in range(B):
        target[b, tgt_len[b] - 1] = 2
        target[b, tgt_len[b]:] = 0
        context[b, ctx_len[b]:] = 0
        frames[b, n_frames[b]:] = 0
        audio[b, n_frames[b]:] = 0
    return dict(frames=frames, audio=audio, context=context, target=target, ctx_len=ctx_len.to(torch.int32),
                tgt_len=tgt_len.to(torch.int32), n_frames=n_frames.to(torch.int32))
