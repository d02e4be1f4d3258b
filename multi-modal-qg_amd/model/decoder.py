"""``model.decoder`` of the reference, re-implemented on the HIP kernels.

Class names, constructor signatures, ``forward`` signatures, return tuples and state-dict keys
follow /root/reference/model/decoder.py (AttnDecoder :49-125, Decoder :7-47) so the reference's
train.py / evaluate.py can import this module unchanged.  Differences, all additive:

* every tensor must live on a ROCm device — there is no CPU path;
* a leading batch dimension is accepted everywhere (``hidden`` [L,B,H], ``word`` with B ids,
  value tensors [B,L,D]); with B == 1 the call is exactly the reference's;
* ``mask_mode``: the reference's three ``pre_soft[n:] = -inf`` statements (decoder.py:79,85,93)
  slice dim 0 of a (1,L) tensor and therefore mask nothing.  Default 0 reproduces that;
  1 applies the intended column masks;
* the per-token debug prints of decoder.py:89,97 are not reproduced.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ._params import LinearParams, LSTMParams, fresh_seed


def _as_batch(t: torch.Tensor, B: int) -> torch.Tensor:
    """(L,D) -> (B,L,D) view (the reference passes one question's rows without a batch dim)."""
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.shape[0] != B:
        if t.shape[0] != 1:
            raise RuntimeError(f"value tensor batch {t.shape[0]} does not match hidden batch {B}")
        t = t.expand(B, *t.shape[1:])
    return t


def _as_lengths(x, B: int, device) -> torch.Tensor:
    if torch.is_tensor(x):
        x = x.reshape(-1).to(device=device, dtype=torch.int32)
        return x if x.numel() == B else x.expand(B).contiguous()
    return torch.full((B,), int(x), device=device, dtype=torch.int32)


class AttnDecoder(nn.Module):
    def __init__(self, num_layers, dropout_p, hidden_dim, n_vocab, word_emb_dim, video_emb_dim, audio_emb_dim,
                 emb_layer, text_max_length, av_max_length, device):
        super().__init__()
        self.num_layers, self.hidden_dim, self.n_vocab, self.dropout_p = num_layers, hidden_dim, n_vocab, dropout_p
        self.text_max_length, self.av_max_length = text_max_length, av_max_length
        self.video_emb_dim, self.audio_emb_dim, self.word_emb_dim = video_emb_dim, audio_emb_dim, word_emb_dim
        self.emb_layer = emb_layer
        self.device = device
        self.mask_mode = 0
        q = word_emb_dim + hidden_dim
        self.text_attn = LinearParams(q, text_max_length)
        self.vid_attn = LinearParams(q, av_max_length)
        self.audio_attn = LinearParams(q, av_max_length)
        self.lstm = LSTMParams(word_emb_dim + hidden_dim + audio_emb_dim + video_emb_dim, hidden_dim, num_layers,
                               dropout=dropout_p)
        self.out_layer = LinearParams(hidden_dim, n_vocab)
        self.initialise_weights()

    def forward(self, word, enc_frames, enc_seq_len, audio_emb, video_emb, hidden, encoder_outputs):
        h, c = hidden
        B = h.shape[1]
        ops.require_device(h, c, audio_emb, video_emb, encoder_outputs, self.emb_layer.weight)
        dev = h.device
        word = word.reshape(-1).to(dev)
        if word.numel() != B:
            raise RuntimeError(f"got {word.numel()} word ids for a hidden state of batch {B}")
        emb = ops.EmbeddingFn.apply(self.emb_layer.weight, word)                       # decoder.py:75
        q = torch.cat((emb, h[-1]), dim=1)                                             # query = [emb | h_top]
        # score segments stacked text | audio | video (concat order of decoder.py:99)
        scores = ops.MultiLinearFn.apply(q, self.text_attn.weight, self.text_attn.bias,
                                         self.audio_attn.weight, self.audio_attn.bias,
                                         self.vid_attn.weight, self.vid_attn.bias)
        text_len = av_len = None
        if self.mask_mode:
            text_len = _as_lengths(enc_seq_len, B, dev)
            av_len = _as_lengths(enc_frames, B, dev)
        attn, ctx = ops.AttentionFn.apply(scores, _as_batch(encoder_outputs, B), _as_batch(audio_emb, B),
                                          _as_batch(video_emb, B), text_len, av_len, self.mask_mode)
        x = torch.cat((emb, ctx), dim=1)                                               # emb | text | audio | video
        y, h_new, c_new = ops.lstm_seq(x.unsqueeze(0), h, c, self.lstm.flat(), self.dropout_p, self.training,
                                       fresh_seed() if (self.training and self.dropout_p > 0) else 0)
        logits = ops.MultiLinearFn.apply(y[0], self.out_layer.weight, self.out_layer.bias)   # decoder.py:106
        Lt, Lav = self.text_max_length, self.av_max_length
        return logits, (h_new, c_new), attn[:, :Lt], attn[:, Lt:Lt + Lav], attn[:, Lt + Lav:]

    def initialise_weights(self):
        self.lstm.reference_init()
        for lin in (self.out_layer, self.text_attn, self.audio_attn, self.vid_attn):
            lin.reference_init()


class Decoder(nn.Module):
    """The older non-attention decoder (decoder.py:7-47): LSTM over [word_emb | av_emb], then a
    Linear to the vocabulary.  Kept for import compatibility; reuses the same kernels."""

    def __init__(self, num_layers, dropout, hidden_dim, n_vocab, word_emb_dim, av_emb_dim, emb_layer):
        super().__init__()
        self.num_layers, self.dropout, self.hidden_dim, self.n_vocab = num_layers, dropout, hidden_dim, n_vocab
        self.word_emb_dim, self.av_emb_dim = word_emb_dim, av_emb_dim
        self.word_embeddings = emb_layer
        self.lstm = LSTMParams(word_emb_dim + av_emb_dim, hidden_dim, num_layers, dropout=dropout)
        self.out_layer = LinearParams(hidden_dim, n_vocab)
        self.initialise_weights()

    def forward(self, text, av_enc_out, hidden):
        n = text.shape[1]
        ops.require_device(av_enc_out, hidden[0])
        emb = ops.EmbeddingFn.apply(self.word_embeddings.weight, text.reshape(-1).to(av_enc_out.device))
        x = torch.cat((emb, av_enc_out.reshape(1, -1).repeat(n, 1)), dim=1)
        y, h, c = ops.lstm_seq(x.unsqueeze(1), hidden[0], hidden[1], self.lstm.flat(), self.dropout, self.training,
                               fresh_seed() if (self.training and self.dropout > 0) else 0)
        logits = ops.MultiLinearFn.apply(y.reshape(n, -1), self.out_layer.weight, self.out_layer.bias)
        return logits.view(n, 1, -1), (h, c)

    def init_state(self, batch_sz):
        dev = self.out_layer.weight.device
        return (torch.zeros(self.num_layers, batch_sz, self.hidden_dim, device=dev),
                torch.zeros(self.num_layers, batch_sz, self.hidden_dim, device=dev))

    def initialise_weights(self):
        self.lstm.reference_init()
        self.out_layer.reference_init()
