"""Pin the CPU oracle against vectors produced by the reference's own classes
(tests/golden/make_golden.py).  fp32, tolerance 1e-4 as BASELINE.json's north_star states
(observed restatement noise is ~1e-6); argmax-decoded token ids are compared bit-exact."""
import numpy as np
import torch

from golden_util import clone_params, collate, load_npz, small_cfg, small_params, small_samples, state_from
from oracle import mmqg_oracle as O
from seeded import decoder_spec, lstm_spec, seeded_params, text_spec

TOL = 1e-4
torch.set_num_threads(4)


from golden_util import close  # noqa: E402  (relative to max|want|, floor 1e-7, logged)


def _grads(dec, text, vid):
    out = {}
    for name, sd in (("vid", vid), ("text", text), ("dec", dec)):
        for k, t in sd.items():
            if t.grad is not None:
                out[f"{name}/{k}"] = t.grad.clone()
    return out


def _run(z, samples, idx, training=True):
    c, cfg = small_cfg(z)
    dec, text, vid = clone_params(*small_params(z))
    for sd in (dec, text, vid):
        for k, t in sd.items():
            if t.is_floating_point() and "running_" not in k:
                t.requires_grad_(True)
    batch = collate([samples[i] for i in idx])
    loss, logits, attn, hidden = O.forward_loss(dec, text, vid, batch, cfg, training=training)
    return c, cfg, (dec, text, vid), batch, loss, logits, attn, hidden


def test_per_question_forward_backward_matches_reference():
    z = load_npz("small_model.npz")
    samples = small_samples(z)
    for b in range(3):
        c, cfg, (dec, text, vid), batch, loss, logits, attn, hidden = _run(z, samples, [b])
        Td = int(batch["tgt_len"][0])
        close(loss, z[f"train/{b}/loss"])
        close(logits[0, :Td], z[f"train/{b}/logits"])
        a = torch.stack([torch.cat([w[0] for w in step]) for step in attn])   # text | audio | video
        close(a, z[f"train/{b}/attn"])
        close(hidden[0][:, 0], z[f"train/{b}/h"][:, 0])
        close(hidden[1][:, 0], z[f"train/{b}/c"][:, 0])
        loss.backward()
        g = _grads(dec, text, vid)
        for k in z.files:
            pre = f"train/{b}/grad/"
            if not k.startswith(pre):
                continue
            name = k[len(pre):]
            if name == "text/word_embeddings.weight":
                name = "dec/emb_layer.weight"        # shared table: one gradient
            close(g[name], z[k])


def test_ragged_batch_equals_mean_of_per_question_gradients():
    z = load_npz("small_model.npz")
    samples = small_samples(z)
    c, cfg, (dec, text, vid), batch, loss, logits, attn, hidden = _run(z, samples, [0, 1, 2])
    want = np.mean([float(z[f"train/{b}/loss"]) for b in range(3)])
    close(loss, np.float32(want))
    for b in range(3):
        Td = int(batch["tgt_len"][b])
        close(logits[b, :Td], z[f"train/{b}/logits"])
        close(hidden[0][:, b], z[f"train/{b}/h"][:, 0])
    loss.backward()
    g = _grads(dec, text, vid)
    for name, t in g.items():
        key = name if name != "dec/emb_layer.weight" else "dec/emb_layer.weight"
        ref = np.mean([z[f"train/{b}/grad/{key}"] for b in range(3)], axis=0)
        close(t, ref)


def test_video_and_text_encoder_outputs():
    z = load_npz("small_model.npz")
    samples = small_samples(z)
    c, cfg = small_cfg(z)
    dec, text, vid = clone_params(*small_params(z))
    batch = collate(samples)
    with torch.no_grad():
        v = O.video_encoder_run(vid, batch["frames"], batch["n_frames"], c["Dv"], c["Lav"], True)
        e, _ = O.text_encoder_run(text, batch["context"], batch["ctx_len"], c["L"], c["H"], c["Lt"])
    for b in range(3):
        close(v[b], z[f"train/{b}/video_emb"])
        close(e[b], z[f"train/{b}/enc_all"])


def test_two_adam_iterations_including_the_twice_stepped_embedding():
    z = load_npz("small_model.npz")
    samples = small_samples(z)
    c, cfg = small_cfg(z)
    dec, text, vid = clone_params(*small_params(z))
    tr = O.OracleTrainer(dec, text, vid, cfg, lr=1e-4)
    for it, b in enumerate((0, 1)):
        loss, _ = tr.step(collate([samples[b]]), training=True)
        close(np.float32(loss), z[f"adam/{it}/loss"])
        for name, sd in (("vid", vid), ("text", text), ("dec", dec)):
            want = state_from(z, f"adam/{it}/{name}")
            for k, t in sd.items():
                if k.endswith("num_batches_tracked"):
                    continue
                # weights move by ~lr per step: compare tightly so a missing second
                # embedding step (2e-4 vs 1e-4) cannot pass
                close(t, want[k], tol=2e-6)


def test_eval_mode_greedy_decode_ids_bit_exact():
    z = load_npz("small_model.npz")
    samples = small_samples(z)
    c, cfg = small_cfg(z)
    dec, text, vid = small_params(z, prefix="adam/1")
    with torch.no_grad():
        for b in range(3):
            batch = collate([samples[b]])
            Td = int(batch["tgt_len"][0])
            ids = O.greedy_decode(dec, text, vid, batch, cfg, Td, stop_at_end=False)
            assert ids[0].tolist() == z[f"eval/{b}/ids"].tolist()
            stop = O.greedy_decode(dec, text, vid, batch, cfg, 8, stop_at_end=True)[0].tolist()
            want = z[f"eval/{b}/ids_stop"].tolist()
            assert stop[:len(want)] == want
            assert all(t == 0 for t in stop[len(want):])
        batch = collate(samples)
        ids = O.greedy_decode(dec, text, vid, batch, cfg, int(batch["tgt_len"].max()), stop_at_end=False)
        for b in range(3):
            n = int(batch["tgt_len"][b])
            assert ids[b, :n].tolist() == z[f"eval/{b}/ids"].tolist()


def test_default_dims_decoder_text_and_feature_bypass():
    z = load_npz("default_dims.npz")
    c = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg/")}
    dec = seeded_params(decoder_spec(c["V"], c["E"], c["H"], c["L"], c["Lt"], c["Lav"], c["Da"], c["Dv"]), c["seed_dec"])
    text = seeded_params(text_spec(c["V"], c["E"], c["H"], c["L"]), c["seed_text"])
    text["word_embeddings.weight"] = dec["emb_layer.weight"]
    vid = seeded_params(lstm_spec("lstm.", c["feat"], c["Dv"], 1), c["seed_vid"])
    g = torch.Generator().manual_seed(c["seed_in"])
    feats = torch.randn(c["T"], c["feat"], generator=g)
    audio = torch.randn(c["T"], c["Da"], generator=g)
    ctx = torch.randint(3, c["V"], (c["ctx"],), generator=g)
    words = torch.randint(3, c["V"], (3,), generator=g)
    nf = torch.tensor([c["T"]])
    with torch.no_grad():
        v = O.video_encoder_run(vid, feats[None], nf, c["Dv"], c["Lav"], False)
        close(v[0, :c["T"]], z["video_emb"])
        enc, hid = O.text_encoder_run(text, ctx[None], torch.tensor([c["ctx"]]), c["L"], c["H"], c["Lt"])
        close(enc[0, :c["ctx"]], z["enc_rows"])
        close(hid[0], z["enc_h"]); close(hid[1], z["enc_c"])
        a = torch.nn.functional.pad(audio, (0, 0, 0, c["Lav"] - c["T"]))[None]
        for i in range(3):
            logits, hid, attn = O.attn_decoder_step(dec, words[i:i + 1], hid, c["L"], enc, a, v,
                                                   torch.tensor([c["ctx"]]), nf)
            close(logits, z[f"dec/{i}/logits"])
            close(torch.cat([w[0] for w in attn]), z[f"dec/{i}/attn"])
            assert int(torch.argmax(logits)) == int(np.argmax(z[f"dec/{i}/logits"]))
        close(hid[0], z["dec_h"]); close(hid[1], z["dec_c"])
