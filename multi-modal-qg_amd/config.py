"""Hyper-parameters and paths under the reference's attribute names (config.py:18-89 of the
reference, including the misspelt ``context_max_lenth``), plus the knobs the batched MI355X
path adds.  ``Config(path)`` overlays a JSON file and ``save_config`` writes one, with the
reference's conventions: every str value except ``optim`` becomes a ``Path`` on load
(config.py:110-111) and paths are stringified on save (config.py:96-97)."""
from __future__ import annotations

import json
import os
from pathlib import Path, PurePath

_RESULTS = Path("results/test/")
_DATASET = Path("dataset")
_DATA = Path("data")
_GLOVE_DIM = 300
_GLOVE = Path("glove.6B")

_DEFAULTS = {
    # results (reference config.py:19-25)
    "output_path": _RESULTS,
    "av_model_path": _RESULTS / "av_model.pth",
    "text_enc_model_path": _RESULTS / "text_enc_model.pth",
    "dec_model_path": _RESULTS / "dec_model.pth",
    "stats_json_path": _RESULTS / "stats.json",
    "stats_pkl_path": _RESULTS / "stats.pkl",
    "learned_weight_path": _RESULTS / "learned_weight.pt",
    # dataset (:29-38)
    "dataset_path": _DATASET,
    "subs_path": _DATASET / "subs",
    "video_path": _DATASET / "vids",
    "audio_path": _DATASET / "audio",
    "salient_text_path": _DATASET / "salient_text",
    "salient_frames_path": _DATASET / "salient_frames",
    "salient_audio_path": _DATASET / "salient_audio_clip",
    "salient_text_file": _DATASET / "salient_text" / "salient_text_list.json",
    "questions_file": _DATASET / "labelled_questions.json",
    "videos_file": _DATASET / "videos.json",
    # data (:41-50)
    "data_path": _DATA,
    "vocab_file": _DATA / "vocab.json",
    "index_to_word_file": _DATA / "index_to_word.json",
    "weights_matrix_file": _DATA / "weight_matrix.npy",
    "preprocessed_text_file": _DATA / "preprocesses_text.json",
    "train_file": _DATA / "train_questions.json",
    "val_file": _DATA / "val_questions.json",
    "test_file": _DATA / "test_questions.json",
    # glove (:53-59)
    "glove_emb_dim": _GLOVE_DIM,
    "glove_path": _GLOVE,
    "glove_file": _GLOVE / f"glove.6B.{_GLOVE_DIM}d.txt",
    "glove_words_file": _GLOVE / f"6B.{_GLOVE_DIM}_words.pkl",
    "glove_idx_file": _GLOVE / f"6B.{_GLOVE_DIM}_idx.pkl",
    "glove_matrix_file": _GLOVE / f"6B.{_GLOVE_DIM}_matrix.npy",
    # hyper-parameters (:62-86)
    "epochs": 100,
    "lr": 1e-04,
    "optim": "adam",
    "audio_emb": 128,
    "av_emb": 128 + 400,
    "vid_mean": [0.43216, 0.394666, 0.37645],
    "vid_std": [0.22803, 0.22145, 0.216989],
    "question_max_length": 21,
    "context_max_lenth": 283,
    "av_max_length": 101,
    "av_in_channels": 3,
    "av_kernel_sz": 3,
    "av_stride": 1,
    "video_hidden_dim": 512,
    "flatten_dim": 1000,
    "text_lstm_hidden_dim": 512,
    "text_lstm_layers": 3,
    "text_lstm_dropout": 0.2,
    "text_non_trainable": False,
    "dec_lstm_hidden_dim": 512,
    "dec_lstm_layers": 3,
    "dec_lstm_dropout": 0.2,
    "best_epoch": None,
    # ---- added by the MI355X build (no reference counterpart) ----
    "batch_size": 64,            # questions per GPU per step
    "attention_mask_mode": 0,    # 0 = reference no-op masks, 1 = intended masks
    "start_token_id": 1,         # <pad>=0, <start>=1, <end>=2 (prepare_data.py:63-66)
    "end_token_id": 2,
    "seed": 0,
}


class Config:
    """Class attributes hold the values (as in the reference, where scripts read
    ``config.lr`` etc. off an instance and ``load_config`` mutates the class)."""

    def __init__(self, config_path=None, make_dirs=True):
        if config_path:
            with open(config_path, "r") as f:
                self.load_config(**json.load(f))
        if make_dirs:
            for d in (self.output_path, self.data_path):
                os.makedirs(d, exist_ok=True)

    @classmethod
    def _public(cls):
        return [k for k, v in vars(Config).items() if not k.startswith("_") and not callable(v)
                and not isinstance(v, (classmethod, staticmethod))]

    def save_config(self):
        out = {}
        for key in self._public():
            val = getattr(Config, key)
            out[key] = str(val) if isinstance(val, PurePath) else val
        with open(self.output_path / "config.json", "w") as f:
            json.dump(out, f)

    def load_config(self, **kwargs):
        known = set(self._public())
        for key, value in kwargs.items():
            if key in known:
                setattr(Config, key, Path(value) if isinstance(value, str) and key != "optim" else value)


for _k, _v in _DEFAULTS.items():
    setattr(Config, _k, _v)
del _k, _v
