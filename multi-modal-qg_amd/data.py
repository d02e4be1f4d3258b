"""Data layer for the batched path: the reference's on-disk formats (utils/dataset.py:8-55,
utils/custom_transforms.py:6-44, prepare_data.py:9-24,59-87) plus what the reference lacks — a
padded, batched collate.

On disk (as the reference's prep scripts write them):
* split file (train/val/test_questions.json): list of {video_id, question_id, context, question, ...};
* vocab.json word -> id with <pad>=0, <start>=1, <end>=2; index_to_word.json str(id) -> word;
* frames  ``v_{video_id}_q_{question_id}_.npy``: uint8 (T,H,W,3) salient frames
  (dataset/get_salient_frames.py:41-46);
* audio   ``v_{video_id}_q_{question_id}_.wav`` — the reference hands the PATH to a torch.hub VGGish
  (unavailable here); this layer additionally accepts ``..._.npy`` (n_clips,128) pre-extracted features.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset


# ---- transforms (names and semantics of utils/custom_transforms.py) -----------------------------
def prepare_sequence(seq: str, to_ix: Dict[str, int]) -> torch.Tensor:
    return torch.tensor([to_ix[w] for w in seq.split()], dtype=torch.long)


class ToFloatTensor:
    """(T,H,W,C) uint8 -> (C,T,H,W) float32 in [0,1]."""

    def __call__(self, vid: torch.Tensor) -> torch.Tensor:
        return vid.permute(3, 0, 1, 2).to(torch.float32) / 255


class Resize:
    """Bicubic resize of the two trailing dims; an int scales the SHORTER side to it."""

    def __init__(self, size):
        self.size = size

    def __call__(self, vid: torch.Tensor) -> torch.Tensor:
        if isinstance(self.size, int):
            scale = float(self.size) / min(vid.shape[-2:])
            return F.interpolate(vid, scale_factor=scale, mode="bicubic", align_corners=False)
        return F.interpolate(vid, size=self.size, mode="bicubic", align_corners=False)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, vid: torch.Tensor) -> torch.Tensor:
        shape = (-1,) + (1,) * (vid.dim() - 1)
        return (vid - torch.as_tensor(self.mean).reshape(shape)) / torch.as_tensor(self.std).reshape(shape)


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


# ---- dataset ------------------------------------------------------------------------------------
class VQGDataset(Dataset):
    """Same constructor and 8-tuple items as the reference's VQGDataset (utils/dataset.py:9,55):
    (frames, audio_file, context_tensor, question_id, question_str, target, context_len, target_len).
    ``audio_file`` is the wav path; ``audio_features(idx)`` loads the .npy features if present."""

    def __init__(self, questions_file, vocab_file, idx_2_word_file, frames_path, audio_path, text_transform=None,
                 video_transform=None):
        with open(questions_file) as f:
            self.questions = json.load(f)
        with open(vocab_file) as f:
            self.vocab = json.load(f)
        with open(idx_2_word_file) as f:
            self.index_to_word = json.load(f)
        self.frames_path, self.audio_path = frames_path, audio_path
        self.text_transform, self.video_transform = text_transform, video_transform

    def __len__(self):
        return len(self.questions)

    def _stem(self, idx):
        q = self.questions[idx]
        return f"v_{q['video_id']}_q_{q['question_id']}_"

    def __getitem__(self, idx):
        q = self.questions[idx]
        context_tensor = self.text_transform(q["context"], self.vocab) if self.text_transform else None
        frames = torch.from_numpy(np.load(os.path.join(self.frames_path, self._stem(idx) + ".npy")))
        if self.video_transform:
            frames = self.video_transform(frames)
        audio_file = os.path.join(self.audio_path, self._stem(idx) + ".wav")
        target = self.text_transform(f"{q['question']} <end>", self.vocab) if self.text_transform else None
        return (frames, audio_file, context_tensor, q["question_id"], q["question"], target,
                context_tensor.shape[0], target.shape[0])

    def audio_features(self, idx) -> Optional[torch.Tensor]:
        p = os.path.join(self.audio_path, self._stem(idx) + ".npy")
        return torch.from_numpy(np.load(p)).to(torch.float32) if os.path.exists(p) else None


def view_frames_like_reference(frames_cthw: torch.Tensor, t_max: int) -> torch.Tensor:
    """(C,T,H,W) -> (t_max,C,H,W): the raw ``view`` of model/encoder.py:64 (memory reinterpretation,
    not a permute) applied per question, then zero padding along the frame axis."""
    Cc, T, hh, ww = frames_cthw.shape
    v = frames_cthw.contiguous().view(T, Cc, hh, ww)
    return F.pad(v, (0, 0, 0, 0, 0, 0, 0, t_max - T))


def collate_questions(items: Sequence[tuple], audio: Sequence[torch.Tensor], n_frames: int, ctx_len: int, tgt_len: int,
                      audio_dim: int = 128, pad_id: int = 0) -> Dict[str, torch.Tensor]:
    """Pad dataset items (+ their (n_clips,audio_dim) audio features) into one batch for
    ``BatchedTrainer``: sequences longer than the trainer's fixed extents are truncated
    (the reference would fail on them too: its attention widths are fixed, config.py:70-71)."""
    B = len(items)
    frames_l, nf = [], []
    for it in items:
        fr = it[0]
        n = min(fr.shape[1], n_frames)
        frames_l.append(view_frames_like_reference(fr[:, :n], n_frames))
        nf.append(n)
    context = torch.full((B, ctx_len), pad_id, dtype=torch.long)
    target = torch.full((B, tgt_len), pad_id, dtype=torch.long)
    aud = torch.zeros(B, n_frames, audio_dim)
    cl, tl = [], []
    for b, it in enumerate(items):
        c, t = it[2][:ctx_len], it[5][:tgt_len]
        context[b, :len(c)] = c
        target[b, :len(t)] = t
        cl.append(len(c)); tl.append(len(t))
        if audio[b] is not None:
            k = min(audio[b].shape[0], nf[b])
            aud[b, :k] = audio[b][:k]
    return dict(frames=torch.stack(frames_l), audio=aud, context=context, target=target,
                ctx_len=torch.tensor(cl, dtype=torch.int32), tgt_len=torch.tensor(tl, dtype=torch.int32),
                n_frames=torch.tensor(nf, dtype=torch.int32),
                question_id=[it[3] for it in items], question=[it[4] for it in items])
