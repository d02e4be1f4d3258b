"""Host-side rows next to the hot path: the reference's on-disk data formats,
the padded collate, and the checkpoint files — CPU only."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture()
def tiny_corpus(tmp_path):
    words = ["<pad>", "<start>", "<end>", "what", "is", "the", "cat", "doing", "a", "sits", "on", "mat"]
    vocab = {w: i for i, w in enumerate(words)}
    (tmp_path / "frames").mkdir(); (tmp_path / "audio").mkdir()
    qs = [{"video_id": "v1", "question_id": 7, "context": "the cat sits on a mat", "question": "what is the cat doing"},
          {"video_id": "v2", "question_id": 9, "context": "a cat", "question": "what is a cat"}]
    rng = np.random.default_rng(0)
    for q, T in zip(qs, (3, 2)):
        stem = f"v_{q['video_id']}_q_{q['question_id']}_"
        np.save(tmp_path / "frames" / (stem + ".npy"), rng.integers(0, 256, (T, 20, 24, 3), dtype=np.uint8))
        np.save(tmp_path / "audio" / (stem + ".npy"), rng.standard_normal((T, 128)).astype(np.float32))
    json.dump(qs, open(tmp_path / "train.json", "w"))
    json.dump(vocab, open(tmp_path / "vocab.json", "w"))
    json.dump({str(i): w for w, i in vocab.items()}, open(tmp_path / "itow.json", "w"))
    return tmp_path, vocab, qs


def test_dataset_items_and_transforms_follow_the_reference(tiny_corpus):
    from utils.custom_transforms import Normalize, Resize, ToFloatTensor, prepare_sequence
    from utils.dataset import VQGDataset
    from mmqg_amd.data import Compose
    root, vocab, qs = tiny_corpus
    ds = VQGDataset(root / "train.json", root / "vocab.json", root / "itow.json", str(root / "frames"), str(root / "audio"),
                    text_transform=prepare_sequence, video_transform=Compose([ToFloatTensor(), Resize(16)]))
    assert len(ds) == 2
    frames, audio_file, ctx, qid, qstr, tgt, cl, tl = ds[0]
    assert frames.shape == (3, 3, 16, 19) and frames.dtype == torch.float32          # (C,T,H,W), short side -> 16
    assert audio_file.endswith("v_v1_q_7_.wav") and qid == 7 and qstr == qs[0]["question"]
    assert ctx.tolist() == [vocab[w] for w in qs[0]["context"].split()] and cl == 6
    assert tgt.tolist()[-1] == vocab["<end>"] and tl == 6
    raw = torch.from_numpy(np.load(root / "frames" / "v_v1_q_7_.npy"))
    f0 = ToFloatTensor()(raw)
    assert f0.shape == (3, 3, 20, 24) and float(f0.max()) <= 1.0
    assert torch.allclose(f0[1, 2], raw[2, :, :, 1].float() / 255)
    n = Normalize([0.5, 0.4, 0.3], [0.2, 0.2, 0.2])(f0)
    assert torch.allclose(n[2], (f0[2] - 0.3) / 0.2)
    assert ds.audio_features(1).shape == (2, 128)


def test_collate_pads_truncates_and_applies_the_reference_view(tiny_corpus):
    from mmqg_amd.data import Compose, Resize, ToFloatTensor, VQGDataset, collate_questions, prepare_sequence
    root, vocab, qs = tiny_corpus
    ds = VQGDataset(root / "train.json", root / "vocab.json", root / "itow.json", str(root / "frames"), str(root / "audio"),
                    text_transform=prepare_sequence, video_transform=Compose([ToFloatTensor(), Resize((8, 8))]))
    items = [ds[0], ds[1]]
    batch = collate_questions(items, [ds.audio_features(0), ds.audio_features(1)], n_frames=4, ctx_len=5, tgt_len=7)
    assert batch["frames"].shape == (2, 4, 3, 8, 8) and batch["audio"].shape == (2, 4, 128)
    assert batch["n_frames"].tolist() == [3, 2] and batch["ctx_len"].tolist() == [5, 2] and batch["tgt_len"].tolist() == [6, 5]
    assert batch["context"][0].tolist() == items[0][2][:5].tolist() and batch["context"][1, 2:].sum() == 0
    # encoder.py:64 views (1,C,T,H,W) memory as (T,C,H,W): same bytes, reinterpreted
    assert torch.equal(batch["frames"][0, :3].reshape(-1), items[0][0].reshape(-1))
    assert float(batch["frames"][1, 2:].abs().sum()) == 0 and float(batch["audio"][1, 2:].abs().sum()) == 0


def test_reference_checkpoint_files_round_trip(tmp_path):
    from config import Config
    from model.decoder import AttnDecoder
    from model.encoder import AudioVideoEncoder, TextEncoder
    from mmqg_amd.checkpoint import load_models, save_models

    class Cfg:
        output_path = tmp_path
        av_model_path = tmp_path / "av_model.pth"
        text_enc_model_path = tmp_path / "text_enc_model.pth"
        dec_model_path = tmp_path / "dec_model.pth"
        learned_weight_path = tmp_path / "learned_weight.pt"

    def make(seed):
        torch.manual_seed(seed)
        emb = torch.nn.Embedding(30, 8)
        return (AudioVideoEncoder(3, 3, 1, 16, 40), TextEncoder(2, 0.1, 16, 8, emb, "cpu"),
                AttnDecoder(2, 0.1, 16, 30, 8, 16, 4, emb, 6, 3, "cpu"))
    a = make(0)
    for last in (False, True):
        save_models(Cfg, *a, last=last)
    names = set(os.listdir(tmp_path))
    assert {"av_model.pth", "text_enc_model.pth", "dec_model.pth", "learned_weight.pt", "last_av_model.pth",
            "last_text_enc.pth", "last_decoder.pth", "last_weigths.pt"} <= names
    sd = torch.load(tmp_path / "dec_model.pth")
    assert "lstm.weight_ih_l0" in sd and "text_attn.weight" in sd and "emb_layer.weight" in sd
    w = torch.load(tmp_path / "learned_weight.pt", weights_only=False)
    assert isinstance(w, torch.nn.Parameter) and torch.equal(w.data, a[2].emb_layer.weight.data)
    b = make(1)
    assert not torch.equal(b[2].out_layer.weight, a[2].out_layer.weight)
    # a reference checkpoint also carries the VGGish weights: they must be skipped, not fail
    av_sd = torch.load(tmp_path / "av_model.pth")
    av_sd["audio_enc.vggish.features.0.weight"] = torch.zeros(3)
    torch.save(av_sd, tmp_path / "av_model.pth")
    load_models(Cfg, *b)
    for ma, mb in zip(a, b):
        for (ka, va), (kb, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
    assert b[1].word_embeddings.weight is b[2].emb_layer.weight
