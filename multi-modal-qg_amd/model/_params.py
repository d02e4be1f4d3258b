"""Parameter containers that reproduce the reference modules' state-dict key names and
initialisation (torch defaults followed by the reference's ``initialise_weights``) without
instantiating torch's compute modules: the arithmetic runs in the HIP kernels."""
from __future__ import annotations

import math
import random

import torch
from torch import nn
from torch.nn import init


class LinearParams(nn.Module):
    """``weight`` [out,in] and ``bias`` [out] — the keys of a torch.nn.Linear."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        init.uniform_(self.bias, -bound, bound)

    def reference_init(self):
        """decoder.py:116-123: Xavier-uniform weight, N(0,1) bias."""
        init.xavier_uniform_(self.weight)
        init.normal_(self.bias)


class LSTMParams(nn.Module):
    """``weight_ih_l{k}``, ``weight_hh_l{k}``, ``bias_ih_l{k}``, ``bias_hh_l{k}`` — the keys and
    layouts of a torch.nn.LSTM ([4H,in] / [4H,H], gate blocks i,f,g,o)."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, dropout: float = 0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.dropout = input_size, hidden_size, num_layers, dropout
        bound = 1 / math.sqrt(hidden_size)
        for l in range(num_layers):
            d = input_size if l == 0 else hidden_size
            for name, shape in ((f"weight_ih_l{l}", (4 * hidden_size, d)), (f"weight_hh_l{l}", (4 * hidden_size, hidden_size)),
                                (f"bias_ih_l{l}", (4 * hidden_size,)), (f"bias_hh_l{l}", (4 * hidden_size,))):
                p = nn.Parameter(torch.empty(*shape))
                init.uniform_(p, -bound, bound)
                self.register_parameter(name, p)

    def reference_init(self):
        """encoder.py:73-78,102-107 / decoder.py:110-114: orthogonal matrices, N(0,1) biases."""
        for p in self.parameters():
            if p.dim() >= 2:
                init.orthogonal_(p.data)
            else:
                init.normal_(p.data)

    def flat(self):
        out = []
        for l in range(self.num_layers):
            out += [getattr(self, f"weight_ih_l{l}"), getattr(self, f"weight_hh_l{l}"),
                    getattr(self, f"bias_ih_l{l}"), getattr(self, f"bias_hh_l{l}")]
        return out


def fresh_seed() -> int:
    """Seed for the dropout streams of one forward call, drawn from torch's CPU generator so
    ``torch.manual_seed`` makes runs repeatable."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())
