"""CPU oracle for the multimodal encoder -> attention-decoder training step.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may call it, and there only as the checker / the timed CPU baseline.

It restates, in plain torch-CPU fp32 arithmetic written in this repository's own
words (explicit gate math, no ``nn.LSTM`` / ``nn.Linear`` modules), what the
reference computes on its hot path:

* LSTM cell, gates ordered i,f,g,o            -> torch ``nn.LSTM`` as used at
  ``model/encoder.py:91`` (text), ``model/encoder.py:54`` (frames),
  ``model/decoder.py:69`` (decoder)
* text encoder step                            -> ``model/encoder.py:95-100``
* frame CNN + LSTM                             -> ``model/encoder.py:58-71``
* attention decoder step                       -> ``model/decoder.py:74-107``
* teacher-forced training step, loss, Adam     -> ``train.py:149-181``

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's
own ``model/decoder.py`` / ``model/encoder.py`` in the build container, drives them
through a re-enactment of ``train.py:149-181`` and stores inputs/outputs as small
``.npz`` fixtures; ``tests/test_oracle_golden.py`` checks this file against them.

Parameters travel as flat ``dict[str, Tensor]`` with the reference's state-dict
key names (``lstm.weight_ih_l0`` ...), one dict per reference module:
``dec`` (AttnDecoder), ``text`` (TextEncoder), ``vid`` (VideoConvLstmEncoder).
The embedding table is shared: ``dec['emb_layer.weight']`` and
``text['word_embeddings.weight']`` are the same tensor (``train.py:236-255``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

MASK_REFERENCE_NOOP = 0   # decoder.py:79,85,93 slice dim 0 of a (1,L) tensor: nothing is masked
MASK_INTENDED = 1         # what the author meant: columns >= valid length get -inf


# --------------------------------------------------------------------------- LSTM
def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor,
              b_ih: Tensor, b_hh: Tensor) -> Tuple[Tensor, Tensor]:
    """One LSTM cell update for a batch.  x (B,In), h/c (B,H); weights in torch layout
    (4H,In)/(4H,H) with row blocks i,f,g,o."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    H = h.shape[1]
    i = torch.sigmoid(gates[:, 0 * H:1 * H])
    f = torch.sigmoid(gates[:, 1 * H:2 * H])
    g = torch.tanh(gates[:, 2 * H:3 * H])
    o = torch.sigmoid(gates[:, 3 * H:4 * H])
    c_new = f * c + i * g
    h_new = o * torch.tanh(c_new)
    return h_new, c_new


def lstm_stack_step(p: Params, prefix: str, num_layers: int, x: Tensor,
                    hidden: Tuple[Tensor, Tensor],
                    drop_masks: Optional[Sequence[Tensor]] = None,
                    active: Optional[Tensor] = None) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    """One time step through ``num_layers`` stacked cells.

    hidden = (h, c) each (L,B,H).  ``drop_masks[l]`` (B,H), already scaled by
    1/(1-p), multiplies the output of layer l before it feeds layer l+1 (torch's
    inter-layer LSTM dropout; never applied to the top layer).  ``active`` (B,) bool:
    rows that are False keep their state (ragged batches: a sample whose sequence
    has ended).  Returns (top-layer output, new hidden).
    """
    h_all, c_all = hidden
    hs, cs = [], []
    inp = x
    for l in range(num_layers):
        h_new, c_new = lstm_cell(inp, h_all[l], c_all[l],
                                 p[f"{prefix}weight_ih_l{l}"], p[f"{prefix}weight_hh_l{l}"],
                                 p[f"{prefix}bias_ih_l{l}"], p[f"{prefix}bias_hh_l{l}"])
        if active is not None:
            keep = active.view(-1, 1)
            h_new = torch.where(keep, h_new, h_all[l])
            c_new = torch.where(keep, c_new, c_all[l])
        hs.append(h_new)
        cs.append(c_new)
        inp = h_new
        if drop_masks is not None and l < num_layers - 1:
            inp = inp * drop_masks[l]
    return hs[-1], (torch.stack(hs), torch.stack(cs))


# ------------------------------------------------------------------- text encoder
def text_encoder_step(text: Params, ids: Tensor, hidden: Tuple[Tensor, Tensor], num_layers: int,
                      drop_masks=None, active=None):
    """``TextEncoder.forward`` for one token per sample (encoder.py:95-100): ids (B,) int64."""
    x = text["word_embeddings.weight"][ids]
    return lstm_stack_step(text, "lstm.", num_layers, x, hidden, drop_masks, active)


def text_encoder_run(text: Params, context: Tensor, ctx_len: Tensor, num_layers: int, hidden_dim: int,
                     text_max_length: int, drop_masks=None):
    """The ``train.py:159-166`` loop, batched: context (B,Tc) ids, ctx_len (B,).
    Returns enc_outputs (B,text_max_length,H) zero-padded, final (h,c) (L,B,H) per sample."""
    B, Tc = context.shape
    dt = text["lstm.weight_ih_l0"].dtype
    h = torch.zeros(num_layers, B, hidden_dim, dtype=dt)
    c = torch.zeros(num_layers, B, hidden_dim, dtype=dt)
    rows: List[Tensor] = []
    for t in range(Tc):
        active = ctx_len > t
        dm = None if drop_masks is None else drop_masks[t]
        out, (h, c) = text_encoder_step(text, context[:, t], (h, c), num_layers, dm, active)
        rows.append(out * active.view(-1, 1).to(dt))
    enc = torch.stack(rows, dim=1) if rows else torch.zeros(B, 0, hidden_dim, dtype=dt)
    enc = F.pad(enc, (0, 0, 0, text_max_length - Tc))
    return enc, (h, c)


# ------------------------------------------------------------------ frame encoder
def _bn_per_question(x: Tensor, w: Tensor, b: Tensor, valid: Tensor, training: bool,
                     running_mean: Optional[Tensor], running_var: Optional[Tensor], eps: float = 1e-5,
                     momentum: float = 0.1):
    """BatchNorm2d whose 'batch' is the frames of ONE question (encoder.py:64 feeds
    (T,C,H,W)).  x (B,T,C,H,W); valid (B,T) marks real frames.  Training mode uses the
    biased variance of that question's frames and moves the running stats once per
    question, in batch order, with the unbiased variance (torch BatchNorm semantics);
    eval mode uses the running stats."""
    if training:
        m = valid.view(*valid.shape, 1, 1, 1).to(x.dtype)
        n = (valid.sum(dim=1).to(x.dtype) * x.shape[3] * x.shape[4]).view(-1, 1, 1, 1, 1)
        mean = (x * m).sum(dim=(1, 3, 4), keepdim=True) / n
        var = (((x - mean) ** 2) * m).sum(dim=(1, 3, 4), keepdim=True) / n
        if running_mean is not None:
            with torch.no_grad():
                for bi in range(x.shape[0]):
                    nb = n[bi].reshape(())
                    running_mean.mul_(1 - momentum).add_(momentum * mean[bi].reshape(-1))
                    running_var.mul_(1 - momentum).add_(momentum * var[bi].reshape(-1) * nb / (nb - 1))
    else:
        mean = running_mean.view(1, 1, -1, 1, 1)
        var = running_var.view(1, 1, -1, 1, 1)
    y = (x - mean) / torch.sqrt(var + eps)
    return y * w.view(1, 1, -1, 1, 1) + b.view(1, 1, -1, 1, 1)


def frame_cnn(vid: Params, frames: Tensor, n_frames: Tensor, training: bool) -> Tensor:
    """conv/ReLU/BN stack of encoder.py:64-67 -> (B,T,flatten).  frames (B,T,C,H,W) is
    the layout AFTER the reference's ``view(T,C,H,W)`` of its (1,C,T,H,W) input
    (encoder.py:64 — a raw reinterpretation, not a permute; ``view_frames_like_reference``
    below applies it per question before questions of different length are padded)."""
    B, T, C, Hh, Ww = frames.shape
    x = frames
    valid = torch.arange(T).view(1, -1) < n_frames.view(-1, 1)

    def block(x, ci, pool):
        Bb, Tt = x.shape[:2]
        y = F.conv2d(x.reshape(Bb * Tt, *x.shape[2:]), vid[f"conv{ci}.weight"], vid[f"conv{ci}.bias"])
        y = torch.relu(y).reshape(Bb, Tt, *y.shape[1:])
        y = _bn_per_question(y, vid[f"bn{ci}.weight"], vid[f"bn{ci}.bias"], valid, training,
                             vid.get(f"bn{ci}.running_mean"), vid.get(f"bn{ci}.running_var"))
        if pool:
            k = vid["conv1.weight"].shape[-1]
            y = F.max_pool2d(y.reshape(Bb * Tt, *y.shape[2:]), k, k).reshape(Bb, Tt, y.shape[2], -1)
            side = int(round(math.sqrt(y.shape[-1])))
            y = y.reshape(Bb, Tt, y.shape[2], side, side)
        return y

    x = block(x, 1, False)
    x = block(x, 2, True)
    x = block(x, 3, False)
    x = block(x, 4, True)
    return x.reshape(B, T, -1)


def view_frames_like_reference(frames_cthw: Tensor, t_max: int) -> Tensor:
    """(C,T,H,W) of one question -> (t_max,C,H,W): the raw ``view`` of encoder.py:64 then zero
    padding along the (new) frame axis."""
    C, T, Hh, Ww = frames_cthw.shape
    v = frames_cthw.contiguous().view(T, C, Hh, Ww)
    return F.pad(v, (0, 0, 0, 0, 0, 0, 0, t_max - T))


def frame_lstm(vid: Params, feats: Tensor, n_frames: Tensor, hidden_dim: int) -> Tensor:
    """Single-layer LSTM over per-frame features (encoder.py:54,69), zero initial state.
    feats (B,T,D) -> (B,T,H); rows past n_frames are zero (train.py:157 pads with zeros)."""
    B, T, _ = feats.shape
    dt = feats.dtype
    h = torch.zeros(1, B, hidden_dim, dtype=dt)
    c = torch.zeros(1, B, hidden_dim, dtype=dt)
    outs = []
    for t in range(T):
        active = n_frames > t
        out, (h, c) = lstm_stack_step(vid, "lstm.", 1, feats[:, t], (h, c), None, active)
        outs.append(out * active.view(-1, 1).to(dt))
    return torch.stack(outs, dim=1)


def video_encoder_run(vid: Params, frames_or_feats: Tensor, n_frames: Tensor, hidden_dim: int,
                      av_max_length: int, training: bool) -> Tensor:
    """VideoConvLstmEncoder.forward + the zero padding of train.py:157 -> (B,av_max_length,H).
    A 5-D input is raw frames (CNN stage runs); a 3-D input (B,T,D) is pre-extracted
    per-frame features that go straight to the LSTM (BASELINE configs 2-5)."""
    feats = frame_cnn(vid, frames_or_feats, n_frames, training) if frames_or_feats.dim() == 5 else frames_or_feats
    out = frame_lstm(vid, feats, n_frames, hidden_dim)
    return F.pad(out, (0, 0, 0, av_max_length - out.shape[1]))


# ------------------------------------------------------------------------ decoder
def _attn_softmax(scores: Tensor, valid_len: Tensor, mask_mode: int) -> Tensor:
    if mask_mode == MASK_INTENDED:
        cols = torch.arange(scores.shape[1]).view(1, -1)
        scores = scores.masked_fill(cols >= valid_len.view(-1, 1), float("-inf"))
    return torch.softmax(scores, dim=1)


def attn_decoder_step(dec: Params, word: Tensor, hidden: Tuple[Tensor, Tensor], num_layers: int,
                      enc_outputs: Tensor, audio_emb: Tensor, video_emb: Tensor,
                      ctx_len: Tensor, n_frames: Tensor, mask_mode: int = MASK_REFERENCE_NOOP,
                      drop_masks=None, active: Optional[Tensor] = None):
    """``AttnDecoder.forward`` (decoder.py:74-107), batched.
    word (B,), hidden (L,B,H)x2, enc_outputs (B,Lt,H), audio_emb (B,Lav,Da), video_emb (B,Lav,Dv).
    Returns logits (B,V), hidden, (text_w (B,Lt), audio_w (B,Lav), video_w (B,Lav)) — the
    reference's return order text, AUDIO, VIDEO (decoder.py:107)."""
    emb = dec["emb_layer.weight"][word]                                   # decoder.py:75
    q = torch.cat((emb, hidden[0][-1]), dim=1)                             # query uses the TOP layer h
    a_t = _attn_softmax(q @ dec["text_attn.weight"].t() + dec["text_attn.bias"], ctx_len, mask_mode)
    c_t = torch.bmm(a_t.unsqueeze(1), enc_outputs).squeeze(1)              # decoder.py:81
    a_v = _attn_softmax(q @ dec["vid_attn.weight"].t() + dec["vid_attn.bias"], n_frames, mask_mode)
    c_v = torch.bmm(a_v.unsqueeze(1), video_emb).squeeze(1)                # decoder.py:87
    a_a = _attn_softmax(q @ dec["audio_attn.weight"].t() + dec["audio_attn.bias"], n_frames, mask_mode)
    c_a = torch.bmm(a_a.unsqueeze(1), audio_emb).squeeze(1)                # decoder.py:95
    x = torch.cat((emb, c_t, c_a, c_v), dim=1)                             # decoder.py:99: emb|text|AUDIO|VIDEO
    top, hidden = lstm_stack_step(dec, "lstm.", num_layers, x, hidden, drop_masks, active)
    logits = top @ dec["out_layer.weight"].t() + dec["out_layer.bias"]     # decoder.py:106
    return logits, hidden, (a_t, a_a, a_v)


# ------------------------------------------------------------------ training step
def forward_loss(dec: Params, text: Params, vid: Params, batch: dict, cfg: dict,
                 training: bool = True, drop: Optional[dict] = None):
    """Batched re-statement of train.py:153-175.  ``batch``: frames (B,T,C,H,W) (see
    ``view_frames_like_reference``) or (B,T,D) features, audio (B,Ta,Da) features, context (B,Tc), ctx_len (B,), target
    (B,Td), tgt_len (B,), n_frames (B,).  Loss = (1/B) sum_b sum_{t<tgt_len[b]}
    CE(logits[b,t], target[b,t]); for B == 1 this is exactly train.py:174's running sum.
    Returns (loss, per-step logits (B,Td,V), attention weights, final hidden)."""
    L, H = cfg["num_layers"], cfg["hidden_dim"]
    Lt, Lav = cfg["text_max_length"], cfg["av_max_length"]
    context, ctx_len = batch["context"], batch["ctx_len"]
    target, tgt_len, n_frames = batch["target"], batch["tgt_len"], batch["n_frames"]
    B, Td = target.shape
    video_emb = video_encoder_run(vid, batch["frames"], n_frames, cfg["video_hidden_dim"], Lav, training)
    audio = batch["audio"]
    audio_emb = F.pad(audio, (0, 0, 0, Lav - audio.shape[1]))             # train.py:156 (decoder contract)
    enc_outputs, hidden = text_encoder_run(text, context, ctx_len, L, H, Lt,
                                           None if drop is None else drop["text"])
    word = torch.full((B,), cfg["start_id"], dtype=torch.long)            # train.py:168
    loss = torch.zeros((), dtype=enc_outputs.dtype)
    all_logits, all_attn = [], []
    for t in range(Td):
        dm = None if drop is None else drop["dec"][t]
        logits, hidden, attn = attn_decoder_step(dec, word, hidden, L, enc_outputs, audio_emb, video_emb,
                                                 ctx_len, n_frames, cfg.get("mask_mode", MASK_REFERENCE_NOOP), dm,
                                                 tgt_len > t)
        ce = F.cross_entropy(logits, target[:, t], reduction="none")
        loss = loss + (ce * (tgt_len > t).to(ce.dtype)).sum() / B         # train.py:174
        word = target[:, t]                                               # teacher forcing, train.py:175
        all_logits.append(logits)
        all_attn.append(attn)
    return loss, torch.stack(all_logits, dim=1), all_attn, hidden


def greedy_decode(dec: Params, text: Params, vid: Params, batch: dict, cfg: dict, max_len: int,
                  stop_at_end: bool = True):
    """validate()/evaluate() greedy decode (train.py:100-110, evaluate.py:70-103), batched,
    eval mode.  Returns token ids (B,max_len) with positions after <end> set to pad (0)."""
    L, H = cfg["num_layers"], cfg["hidden_dim"]
    Lt, Lav = cfg["text_max_length"], cfg["av_max_length"]
    n_frames = batch["n_frames"]
    B = batch["context"].shape[0]
    video_emb = video_encoder_run(vid, batch["frames"], n_frames, cfg["video_hidden_dim"], Lav, False)
    audio_emb = F.pad(batch["audio"], (0, 0, 0, Lav - batch["audio"].shape[1]))
    enc_outputs, hidden = text_encoder_run(text, batch["context"], batch["ctx_len"], L, H, Lt, None)
    word = torch.full((B,), cfg["start_id"], dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    out = torch.zeros(B, max_len, dtype=torch.long)
    for t in range(max_len):
        logits, hidden, _ = attn_decoder_step(dec, word, hidden, L, enc_outputs, audio_emb, video_emb,
                                              batch["ctx_len"], n_frames, cfg.get("mask_mode", 0), None)
        word = torch.argmax(logits, dim=1)
        out[:, t] = torch.where(done, torch.zeros_like(word), word)
        if stop_at_end:
            done = done | (word == cfg["end_id"])
    return out


def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-4,
                b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam (defaults, no weight decay / amsgrad) as used at train.py:265-267, in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class OracleTrainer:
    """train.py:149-181 around the functions above: zero_grad, forward, backward via torch
    autograd on the restatement, three Adam optimizers in the order av, text, dec.  The
    embedding table belongs to BOTH the text and the decoder optimizer (train.py:236,
    245,255,266-267) so it is stepped twice per iteration, each with its own moments."""

    def __init__(self, dec: Params, text: Params, vid: Params, cfg: dict, lr: float = 1e-4):
        self.dec, self.text, self.vid, self.cfg, self.lr = dec, text, vid, cfg, lr
        assert text["word_embeddings.weight"] is dec["emb_layer.weight"]
        self.step_no = 0
        self._groups = []
        for name, sd in (("vid", vid), ("text", text), ("dec", dec)):
            ps = [(k, t) for k, t in sd.items() if t.is_floating_point() and "running_" not in k]
            self._groups.append((name, ps, {k: (torch.zeros_like(t), torch.zeros_like(t)) for k, t in ps}))

    def trainable(self):
        seen, out = set(), []
        for _, ps, _ in self._groups:
            for _, t in ps:
                if id(t) not in seen:
                    seen.add(id(t))
                    out.append(t)
        return out

    def step(self, batch: dict, training: bool = True, drop: Optional[dict] = None):
        for t in self.trainable():
            t.requires_grad_(True)
            t.grad = None
        loss, logits, _, _ = forward_loss(self.dec, self.text, self.vid, batch, self.cfg, training, drop)
        loss.backward()
        self.step_no += 1
        with torch.no_grad():
            for _, ps, state in self._groups:
                for k, t in ps:
                    if t.grad is None:
                        continue
                    m, v = state[k]
                    adam_update(t, t.grad, m, v, self.step_no, self.lr)
        return float(loss.detach()), logits.detach()


# --------------------------------------------------------------------------- BLEU
def sentence_bleu_charwise(reference_words: Sequence[str], hypothesis: Sequence[str],
                           weights=(0.25, 0.25, 0.25, 0.25)) -> float:
    """nltk.translate.bleu_score.sentence_bleu (nltk 3.x, no smoothing) as the reference
    calls it (train.py:115-119): the list of reference *words* is passed as the list of
    references, so every 'reference' is one word string that nltk iterates character by
    character, while the hypothesis is a list of word strings."""
    from collections import Counter
    from fractions import Fraction

    refs = [list(w) for w in reference_words]
    hyp = list(hypothesis)

    def ngrams(seq, n):
        return [tuple(seq[i:i + n]) for i in range(len(seq) - n + 1)]

    p_n = []
    for n in range(1, len(weights) + 1):
        counts = Counter(ngrams(hyp, n)) if len(hyp) >= n else Counter()
        max_counts: Dict[tuple, int] = {}
        for r in refs:
            rc = Counter(ngrams(r, n)) if len(r) >= n else Counter()
            for g in counts:
                max_counts[g] = max(max_counts.get(g, 0), rc[g])
        clipped = {g: min(c, max_counts[g]) for g, c in counts.items()}
        p_n.append(Fraction(sum(clipped.values()), max(1, sum(counts.values())), _normalize=False))
    if p_n[0].numerator == 0:
        return 0.0
    hyp_len = len(hyp)
    if hyp_len == 0:
        return 0.0
    ref_len = min((len(r) for r in refs), key=lambda rl: (abs(rl - hyp_len), rl)) if refs else 0
    bp = 1.0 if hyp_len > ref_len else math.exp(1 - ref_len / hyp_len)
    s = 0.0
    for w, p in zip(weights, p_n):
        if p.numerator == 0:
            # nltk method0 smoothing: replace a zero precision by sys.float_info.min
            import sys
            s += w * math.log(sys.float_info.min)
        else:
            s += w * math.log(p.numerator / p.denominator)
    return bp * math.exp(s)
