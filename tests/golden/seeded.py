"""Deterministic parameter / input generators shared by ``make_golden.py`` (which feeds
them to the reference) and by the tests (which feed them to the oracle and the HIP path).
Large tensors are never committed: both sides regenerate them from the seed with the
same torch CPU generator calls, in the same order."""
from __future__ import annotations

from collections import OrderedDict

import torch


def _fill(g: torch.Generator, shape, scale: float) -> torch.Tensor:
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def lstm_spec(prefix: str, in_dim: int, hidden: int, layers: int):
    spec = []
    for l in range(layers):
        d = in_dim if l == 0 else hidden
        spec += [(f"{prefix}weight_ih_l{l}", (4 * hidden, d), d ** -0.5),
                 (f"{prefix}weight_hh_l{l}", (4 * hidden, hidden), hidden ** -0.5),
                 (f"{prefix}bias_ih_l{l}", (4 * hidden,), 0.3),
                 (f"{prefix}bias_hh_l{l}", (4 * hidden,), 0.3)]
    return spec


def decoder_spec(V, E, H, L, Lt, Lav, Da, Dv):
    q = E + H
    spec = [("emb_layer.weight", (V, E), 0.6),
            ("text_attn.weight", (Lt, q), q ** -0.5), ("text_attn.bias", (Lt,), 0.5),
            ("vid_attn.weight", (Lav, q), q ** -0.5), ("vid_attn.bias", (Lav,), 0.5),
            ("audio_attn.weight", (Lav, q), q ** -0.5), ("audio_attn.bias", (Lav,), 0.5)]
    spec += lstm_spec("lstm.", E + H + Da + Dv, H, L)
    spec += [("out_layer.weight", (V, H), H ** -0.5), ("out_layer.bias", (V,), 0.3)]
    return spec


def text_spec(V, E, H, L):
    return [("word_embeddings.weight", (V, E), 0.6)] + lstm_spec("lstm.", E, H, L)


def seeded_params(spec, seed: int) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    return OrderedDict((name, _fill(g, shape, scale)) for name, shape, scale in spec)
