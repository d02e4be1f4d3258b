#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE's own classes.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``model/decoder.py`` and ``model/encoder.py`` from /root/reference unchanged
(``torchvision`` is absent here and only needed by the out-of-scope VideoResnetEncoder, so
an empty module object is registered under that name before the import), drives them the
way ``train.py:149-181`` / ``train.py:87-110`` do — one question at a time, audio injected as
an (n_frames, audio_emb_dim) feature tensor because AudioEncoder needs a remote
torch.hub fetch — and writes inputs + outputs as .npz.  Only data is stored here.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from seeded import decoder_spec, lstm_spec, seeded_params, text_spec  # noqa: E402

REF = "/root/reference"
sys.path.insert(0, REF)
_tv = types.ModuleType("torchvision")
_tv.models = types.ModuleType("torchvision.models")
sys.modules.setdefault("torchvision", _tv)
sys.modules.setdefault("torchvision.models", _tv.models)
from model.decoder import AttnDecoder, Decoder  # noqa: E402
from model.encoder import TextEncoder, VideoConvLstmEncoder  # noqa: E402

torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())   # decoder.py:89,97 print every token


def npz_state(prefix, sd):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in sd.items()}


def run_question(av, text_enc, dec, sample, cfg, teacher_forcing=True, max_len=None, stop_at_end=False):
    """One question through the reference modules (train.py:153-175 / validate train.py:81-110)."""
    frames, audio, ctx, tgt = sample["frames"], sample["audio"], sample["context"], sample["target"]
    video_emb = av(frames.unsqueeze(0)).squeeze(1)                 # (T,1,H) -> (T,H)
    n_frames = video_emb.shape[0]
    pad_a = F.pad(audio, (0, 0, 0, cfg["Lav"] - n_frames))
    pad_v = F.pad(video_emb, (0, 0, 0, cfg["Lav"] - n_frames))
    hid = text_enc.init_state(1)
    enc_all = torch.zeros(cfg["Lt"], text_enc.hidden_dim)
    for ei in range(len(ctx)):
        out, hid = text_enc(ctx[ei], hid)
        enc_all[ei] = out[0, 0]
    word = torch.tensor([[cfg["start_id"]]])
    loss = 0
    logits_all, attn_all, ids = [], [], []
    steps = len(tgt) if max_len is None else max_len
    with quiet():
        for di in range(steps):
            logits, hid, a_t, a_a, a_v = dec(word, n_frames, torch.tensor([len(ctx)]), pad_a, pad_v, hid, enc_all)
            logits_all.append(logits[0])
            attn_all.append(torch.cat((a_t[0], a_a[0], a_v[0])))
            if di < len(tgt):
                loss = loss + F.cross_entropy(logits, tgt[di].view(-1))
            if teacher_forcing:
                word = tgt[di]
            else:
                nxt = torch.argmax(F.softmax(logits, dim=1), dim=1, keepdim=True)
                ids.append(int(nxt))
                word = nxt.detach()
                if stop_at_end and ids[-1] == cfg["end_id"]:
                    break
    return dict(loss=loss, logits=torch.stack(logits_all), attn=torch.stack(attn_all), hidden=hid,
                video_emb=pad_v, enc_all=enc_all, ids=ids)


def make_small():
    cfg = dict(V=50, E=12, H=16, L=3, Lt=9, Lav=5, Da=6, Dv=16, img=40, flatten=40, start_id=1, end_id=2)
    torch.manual_seed(1234)
    emb = torch.nn.Embedding(cfg["V"], cfg["E"])
    with torch.no_grad():
        emb.weight.copy_(torch.randn(cfg["V"], cfg["E"]) * 0.6)
    av = VideoConvLstmEncoder(3, 3, 1, cfg["Dv"], cfg["flatten"])
    text_enc = TextEncoder(cfg["L"], 0.0, cfg["H"], cfg["E"], emb, "cpu")
    dec = AttnDecoder(cfg["L"], 0.0, cfg["H"], cfg["V"], cfg["E"], cfg["Dv"], cfg["Da"], emb, cfg["Lt"], cfg["Lav"], "cpu")
    with torch.no_grad():   # non-trivial BN affine so its gradient path is exercised
        for i in (1, 2, 3, 4):
            bn = getattr(av, f"bn{i}")
            bn.weight.copy_(1 + 0.2 * torch.randn_like(bn.weight))
            bn.bias.copy_(0.1 * torch.randn_like(bn.bias))
        # keep the N(0,1) LSTM biases of the reference init but scale the attention biases
        # down a little so the softmaxes are not one-hot
    n_frames = [3, 2, 3]
    ctx_len = [5, 7, 4]
    tgt_len = [4, 6, 5]
    samples = []
    for b in range(3):
        tgt = torch.randint(3, cfg["V"], (tgt_len[b],))
        tgt[-1] = cfg["end_id"]
        samples.append(dict(frames=torch.rand(3, n_frames[b], cfg["img"], cfg["img"]),
                            audio=torch.randn(n_frames[b], cfg["Da"]),
                            context=torch.randint(3, cfg["V"], (ctx_len[b],)),
                            target=tgt))
    out = {"cfg/" + k: np.array(v) for k, v in cfg.items()}
    out.update(npz_state("init/vid", av.state_dict()))
    out.update(npz_state("init/text", text_enc.state_dict()))
    out.update(npz_state("init/dec", dec.state_dict()))
    for b, s in enumerate(samples):
        for k, v in s.items():
            out[f"in/{b}/{k}"] = v.numpy()

    # ---- (1) train-mode forward/backward of every question from the SAME initial weights
    av.train(); text_enc.train(); dec.train()
    init_bn = {k: v.clone() for k, v in av.state_dict().items()}
    for b, s in enumerate(samples):
        av.load_state_dict(init_bn)
        for m in (av, text_enc, dec):
            m.zero_grad()
        r = run_question(av, text_enc, dec, s, cfg)
        r["loss"].backward()
        out[f"train/{b}/loss"] = r["loss"].detach().numpy()
        out[f"train/{b}/logits"] = r["logits"].detach().numpy()
        out[f"train/{b}/attn"] = r["attn"].detach().numpy()
        out[f"train/{b}/h"] = r["hidden"][0].detach().numpy()
        out[f"train/{b}/c"] = r["hidden"][1].detach().numpy()
        out[f"train/{b}/video_emb"] = r["video_emb"].detach().numpy()
        out[f"train/{b}/enc_all"] = r["enc_all"].detach().numpy()
        for name, mod in (("vid", av), ("text", text_enc), ("dec", dec)):
            for k, p in mod.named_parameters():
                out[f"train/{b}/grad/{name}/{k}"] = p.grad.detach().numpy()
    av.load_state_dict(init_bn)

    # ---- (2) two optimizer iterations exactly as train.py:149-181 (questions 0 then 1)
    opt_av = torch.optim.Adam(av.parameters(), lr=1e-4)
    opt_text = torch.optim.Adam(text_enc.parameters(), lr=1e-4)
    opt_dec = torch.optim.Adam(dec.parameters(), lr=1e-4)
    for it, b in enumerate((0, 1)):
        opt_av.zero_grad(); opt_text.zero_grad(); opt_dec.zero_grad()
        r = run_question(av, text_enc, dec, samples[b], cfg)
        r["loss"].backward()
        opt_av.step(); opt_text.step(); opt_dec.step()
        out[f"adam/{it}/loss"] = r["loss"].detach().numpy()
        out.update(npz_state(f"adam/{it}/vid", av.state_dict()))
        out.update(npz_state(f"adam/{it}/text", text_enc.state_dict()))
        out.update(npz_state(f"adam/{it}/dec", dec.state_dict()))

    # ---- (3) eval mode (BN running stats, as left by the iterations above): validate-style
    # greedy decode over target_len steps (train.py:100-110) and evaluate-style decode that
    # stops at <end> or question_max_length (evaluate.py:70-103)
    av.eval(); text_enc.eval(); dec.eval()
    with torch.no_grad():
        for b, s in enumerate(samples):
            r = run_question(av, text_enc, dec, s, cfg, teacher_forcing=False)
            out[f"eval/{b}/loss"] = r["loss"].numpy()
            out[f"eval/{b}/logits"] = r["logits"].numpy()
            out[f"eval/{b}/ids"] = np.array(r["ids"], dtype=np.int64)
            r2 = run_question(av, text_enc, dec, s, cfg, teacher_forcing=False, max_len=8, stop_at_end=True)
            out[f"eval/{b}/ids_stop"] = np.array(r2["ids"], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "small_model.npz"), **out)
    print("small_model.npz:", len(out), "arrays")


def make_plain_decoder():
    """The older non-attention ``Decoder`` (decoder.py:7-47): a 5-token teacher-forced call, eval mode
    (dropout off), forward outputs and every parameter gradient of sum(logits * probe)."""
    torch.manual_seed(21)
    V, E, Dav, H, L, n = 37, 12, 10, 16, 2, 5
    emb = torch.nn.Embedding(V, E)
    dec = Decoder(L, 0.3, H, V, E, Dav, emb).eval()
    text = torch.randint(0, V, (1, n))
    av = torch.randn(1, Dav)
    h0, c0 = torch.randn(L, 1, H) * 0.3, torch.randn(L, 1, H) * 0.3
    probe = torch.randn(n, 1, V)
    logits, (h, c) = dec(text, av, (h0, c0))
    ((logits * probe).sum() + (h * 0.5).sum() + (c * 0.25).sum()).backward()
    out = dict(text=text.numpy(), av=av.numpy(), h0=h0.numpy(), c0=c0.numpy(), probe=probe.numpy(),
               logits=logits.detach().numpy().copy(), h=h.detach().numpy().copy(), c=c.detach().numpy().copy(),
               dims=np.array([V, E, Dav, H, L, n]))
    out.update(npz_state("sd", dec.state_dict()))
    for k, p in dec.named_parameters():
        out[f"grad/{k}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "plain_decoder.npz"), **out)
    print("plain_decoder.npz", len(out), "arrays")


def make_default_dims():
    """config.py default widths (E=300,H=512,L=3,Lt=283,Lav=101,Da=128,Dv=512), tiny vocab.
    Weights come from ``seeded_params`` so only seeds + outputs are stored."""
    cfg = dict(V=120, E=300, H=512, L=3, Lt=283, Lav=101, Da=128, Dv=512, seed_dec=11, seed_text=12,
               seed_vid=13, seed_in=14, feat=2048, T=4, ctx=6)
    dec_sd = seeded_params(decoder_spec(cfg["V"], cfg["E"], cfg["H"], cfg["L"], cfg["Lt"], cfg["Lav"], cfg["Da"], cfg["Dv"]), cfg["seed_dec"])
    text_sd = seeded_params(text_spec(cfg["V"], cfg["E"], cfg["H"], cfg["L"]), cfg["seed_text"])
    text_sd["word_embeddings.weight"] = dec_sd["emb_layer.weight"]
    vid_sd = seeded_params(lstm_spec("lstm.", cfg["feat"], cfg["Dv"], 1), cfg["seed_vid"])
    emb = torch.nn.Embedding(cfg["V"], cfg["E"])
    text_enc = TextEncoder(cfg["L"], 0.2, cfg["H"], cfg["E"], emb, "cpu")
    dec = AttnDecoder(cfg["L"], 0.2, cfg["H"], cfg["V"], cfg["E"], cfg["Dv"], cfg["Da"], emb, cfg["Lt"], cfg["Lav"], "cpu")
    dec.load_state_dict(dec_sd)
    text_enc.load_state_dict(text_sd)
    frame_lstm = torch.nn.LSTM(cfg["feat"], cfg["Dv"])   # the LSTM stage of encoder.py:54,69 fed features
    frame_lstm.load_state_dict({k[len("lstm."):]: v for k, v in vid_sd.items()})
    text_enc.eval(); dec.eval()
    g = torch.Generator().manual_seed(cfg["seed_in"])
    feats = torch.randn(cfg["T"], cfg["feat"], generator=g)
    audio = torch.randn(cfg["T"], cfg["Da"], generator=g)
    ctx = torch.randint(3, cfg["V"], (cfg["ctx"],), generator=g)
    words = torch.randint(3, cfg["V"], (3,), generator=g)
    out = {"cfg/" + k: np.array(v) for k, v in cfg.items()}
    with torch.no_grad():
        video_emb = frame_lstm(feats.view(cfg["T"], 1, -1))[0].squeeze(1)
        pad_v = F.pad(video_emb, (0, 0, 0, cfg["Lav"] - cfg["T"]))
        pad_a = F.pad(audio, (0, 0, 0, cfg["Lav"] - cfg["T"]))
        hid = text_enc.init_state(1)
        enc_all = torch.zeros(cfg["Lt"], cfg["H"])
        for ei in range(cfg["ctx"]):
            o, hid = text_enc(ctx[ei], hid)
            enc_all[ei] = o[0, 0]
        out["video_emb"] = video_emb.numpy()
        out["enc_rows"] = enc_all[:cfg["ctx"]].numpy()
        out["enc_h"] = hid[0].numpy(); out["enc_c"] = hid[1].numpy()
        with quiet():
            for i in range(3):
                logits, hid, a_t, a_a, a_v = dec(words[i], cfg["T"], torch.tensor([cfg["ctx"]]), pad_a, pad_v, hid, enc_all)
                out[f"dec/{i}/logits"] = logits.numpy()
                out[f"dec/{i}/attn"] = torch.cat((a_t[0], a_a[0], a_v[0])).numpy()
        out["dec_h"] = hid[0].numpy(); out["dec_c"] = hid[1].numpy()
    np.savez_compressed(os.path.join(HERE, "default_dims.npz"), **out)
    print("default_dims.npz:", len(out), "arrays")


def _ref_file_module(name, rel):
    """A reference source file imported from where it lies (the build repo has a ``utils`` package of the
    same name, so the file is loaded by path instead of through ``import utils``)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_data_layer():
    """The reference's data layer on a tiny corpus (utils/dataset.py:8-55, utils/custom_transforms.py:6-44):
    the corpus itself (split / vocab / index_to_word JSON text, uint8 frame arrays) and, for every question,
    the 8-tuple ``VQGDataset.__getitem__`` returns under two transform stacks — train.py:229's
    [ToFloatTensor, Resize(int)] and evaluate.py:163's [ToFloatTensor, Resize((h,w)), Normalize]."""
    import json
    import tempfile
    ds_mod = _ref_file_module("ref_utils_dataset", "utils/dataset.py")
    tf = _ref_file_module("ref_utils_custom_transforms", "utils/custom_transforms.py")

    class Stack:                      # torchvision.transforms.Compose is what train.py:229 uses; it is absent here
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    words = ["<pad>", "<start>", "<end>", "what", "is", "the", "cat", "doing", "a", "sits", "on", "mat", "why", "dog",
             "runs", "fast"]
    vocab = {w: i for i, w in enumerate(words)}
    qs = [{"video_id": "abc", "question_id": 7, "context": "the cat sits on a mat", "question": "what is the cat doing"},
          {"video_id": "d-9", "question_id": 12, "context": "a dog runs", "question": "why is the dog fast"},
          {"video_id": "abc", "question_id": 3, "context": "a cat", "question": "what is a cat"}]
    shapes = [(3, 20, 24), (2, 30, 18), (4, 16, 16)]       # (T, H, W): landscape, portrait, square
    rng = np.random.default_rng(20260203)
    out = {"json/questions": np.array(json.dumps(qs)), "json/vocab": np.array(json.dumps(vocab)),
           "json/index_to_word": np.array(json.dumps({str(i): w for w, i in vocab.items()})),
           "cfg/resize_int": np.array(14), "cfg/resize_hw": np.array([12, 10]),
           "cfg/mean": np.array([0.43216, 0.394666, 0.37645], dtype=np.float32),      # config.py vid_mean / vid_std
           "cfg/std": np.array([0.22803, 0.22145, 0.216989], dtype=np.float32)}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "frames")); os.makedirs(os.path.join(d, "audio"))
        for i, (q, (T, hh, ww)) in enumerate(zip(qs, shapes)):
            arr = rng.integers(0, 256, (T, hh, ww, 3), dtype=np.uint8)
            out[f"raw/{i}/frames"] = arr
            np.save(os.path.join(d, "frames", f"v_{q['video_id']}_q_{q['question_id']}_.npy"), arr)
        for name, obj in (("q.json", qs), ("vocab.json", vocab), ("itow.json", {str(i): w for w, i in vocab.items()})):
            json.dump(obj, open(os.path.join(d, name), "w"))
        stacks = {"train": Stack([tf.ToFloatTensor(), tf.Resize(int(out["cfg/resize_int"]))]),
                  "eval": Stack([tf.ToFloatTensor(), tf.Resize(tuple(int(v) for v in out["cfg/resize_hw"])),
                                 tf.Normalize(out["cfg/mean"].tolist(), out["cfg/std"].tolist())])}
        for sname, stack in stacks.items():
            ds = ds_mod.VQGDataset(os.path.join(d, "q.json"), os.path.join(d, "vocab.json"), os.path.join(d, "itow.json"),
                                   os.path.join(d, "frames"), os.path.join(d, "audio"), tf.prepare_sequence, stack)
            out[f"{sname}/len"] = np.array(len(ds))
            for i in range(len(ds)):
                frames, audio_file, ctx, qid, qstr, tgt, cl, tl = ds[i]
                out[f"{sname}/{i}/frames"] = frames.numpy()
                out[f"{sname}/{i}/audio_file"] = np.array(os.path.relpath(audio_file, d))
                out[f"{sname}/{i}/context"] = ctx.numpy()
                out[f"{sname}/{i}/question_id"] = np.array(qid)
                out[f"{sname}/{i}/question"] = np.array(qstr)
                out[f"{sname}/{i}/target"] = tgt.numpy()
                out[f"{sname}/{i}/lens"] = np.array([cl, tl])
    np.savez_compressed(os.path.join(HERE, "data_layer.npz"), **out)
    print("data_layer.npz:", len(out), "arrays")


if __name__ == "__main__":
    make_small()
    make_default_dims()
    make_plain_decoder()
    make_data_layer()
