"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, the ctypes structures match the C layout, config / module host logic."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_mod():
    import mmqg_amd  # noqa: F401
    from mmqg_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_library_exports_every_symbol_declared_in_the_header(lib_mod):
    header = open(os.path.join(ROOT, "include", "mmqg.h")).read()
    declared = set(re.findall(r"\b(mmqg_[a-z0-9_]+)\s*\(", header))
    declared -= {"mmqg_stream"}
    lib = C.CDLL(lib_mod.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mmqg.h but not exported"
    assert declared == set(lib_mod.SIGNATURES), (declared ^ set(lib_mod.SIGNATURES))
    assert lib.mmqg_abi_version() == lib_mod.ABI_VERSION


def test_ctypes_structs_match_the_c_layout(lib_mod, tmp_path):
    """Compile a tiny C program against include/mmqg.h that prints sizeof/offsetof and compare
    with the ctypes mirror."""
    fields = {"mmqg_attn_values": ("AttnValues", ["B", "text", "video_stride_b", "mask_mode", "zero_past_len"]),
              "mmqg_lstm_seq": ("LstmSeq", ["x", "w_hh", "w_hhT", "w_ihT", "lens", "seed", "seed_offset", "gates", "y_stride_b",
                                            "persist_ws", "persist_ws_bytes"]),
              "mmqg_lstm_seq_grad": ("LstmSeqGrad", ["dy", "dgates", "lddx", "db_hh", "dc0", "phase"]),
              "mmqg_decoder_seq": ("DecoderSeq", ["values", "xemb", "b_hh", "w_ihT", "w_attn_hT", "seed_offset", "scores",
                                                  "ld_attn", "hdrop", "phase", "h0_stride_l", "attn_ws", "persist_ws", "persist_ws_bytes"]),
              "mmqg_decoder_seq_grad": ("DecoderSeqGrad", ["dhtop", "ld_ds", "dxemb", "db_hh", "n_text_rows", "dvideo_stride_b", "phase", "dh_pre"]),
              "mmqg_decoder_decode": ("DecoderDecode", ["values", "emb_table", "b_hh", "w_out", "start_id", "seed", "target",
                                                        "ids", "ld_attn", "xemb", "hs", "logits", "keep_logits"]),
              "mmqg_transpose_job": ("TransposeJob", ["src", "ld_src", "rows", "cols", "dst", "ld_dst"]),
              "mmqg_gemm_problem": ("GemmProblem", ["M", "K", "A", "lda", "B", "C", "ldc", "beta"]),
              "mmqg_batch_pack": ("BatchPack", ["B", "audio_rows", "frame_inner", "frames", "n_frames", "start_id", "feats",
                                                "audio_stride_b", "row_w", "n_frames_out"]),
              "mmqg_cnn_block": ("CnnBlock", ["cout", "pool", "w", "running_var", "argmax", "stats", "shift"]),
              "mmqg_frame_cnn": ("FrameCnn", ["B", "training", "time_major", "eps", "momentum", "frames", "n_frames", "block"]),
              "mmqg_frame_cnn_grad": ("FrameCnnGrad", ["dfeat", "dz", "dw", "dbias", "dgamma", "dbeta"])}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mmqg.h"', 'int main(void){']
    for cname, (_, fs) in fields.items():
        src.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fs:
            src.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    src.append("return 0;}")
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, (pyname, fs) in fields.items():
        cls = getattr(lib_mod, pyname)
        assert int(out[cname]) == C.sizeof(cls), cname
        for f in fs:
            assert int(out[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_cpu_tensors_are_rejected_not_silently_computed(lib_mod):
    from mmqg_amd import ops
    from model.decoder import AttnDecoder
    from model.encoder import TextEncoder
    emb = torch.nn.Embedding(20, 8)
    dec = AttnDecoder(2, 0.0, 8, 20, 8, 8, 4, emb, 5, 3, "cpu")
    text = TextEncoder(2, 0.0, 8, 8, emb, "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(torch.tensor([[1]]), 2, torch.tensor([3]), torch.zeros(3, 4), torch.zeros(3, 8),
            (torch.zeros(2, 1, 8), torch.zeros(2, 1, 8)), torch.zeros(5, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        text(torch.tensor(1), text.init_state(1))
    with pytest.raises(RuntimeError):
        ops.linear_fwd(torch.zeros(2, 2), torch.zeros(2, 2), None)


def test_state_dict_keys_and_shapes_follow_the_reference():
    from model.decoder import AttnDecoder, Decoder
    from model.encoder import AudioVideoEncoder, TextEncoder
    emb = torch.nn.Embedding(50, 12)
    dec = AttnDecoder(3, 0.2, 16, 50, 12, 16, 6, emb, 9, 5, "cpu")
    keys = list(dec.state_dict().keys())
    assert keys[:7] == ["emb_layer.weight", "text_attn.weight", "text_attn.bias", "vid_attn.weight", "vid_attn.bias",
                        "audio_attn.weight", "audio_attn.bias"]
    assert keys[-2:] == ["out_layer.weight", "out_layer.bias"]
    assert dec.lstm.weight_ih_l0.shape == (64, 12 + 16 + 6 + 16) and dec.lstm.weight_hh_l2.shape == (64, 16)
    assert dec.text_attn.weight.shape == (9, 28) and dec.vid_attn.weight.shape == (5, 28)
    text = TextEncoder(3, 0.2, 16, 12, emb, "cpu")
    assert list(text.state_dict().keys())[0] == "word_embeddings.weight"
    assert text.word_embeddings.weight is dec.emb_layer.weight
    av = AudioVideoEncoder(3, 3, 1, 16, 40)
    k = set(av.state_dict().keys())
    assert {"video_enc.conv1.weight", "video_enc.bn4.running_var", "video_enc.bn1.num_batches_tracked",
            "video_enc.lstm.weight_ih_l0", "video_enc.lstm.bias_hh_l0"} <= k
    old = Decoder(2, 0.1, 16, 50, 12, 8, emb)
    assert "lstm.weight_ih_l1" in old.state_dict() and old.lstm.weight_ih_l0.shape == (64, 20)
    h, c = text.init_state(3)
    assert h.shape == (3, 3, 16) and float(h.abs().sum()) == 0


def test_reference_initialisation_scheme():
    from model.decoder import AttnDecoder
    torch.manual_seed(0)
    emb = torch.nn.Embedding(30, 8)
    dec = AttnDecoder(2, 0.0, 32, 30, 8, 32, 4, emb, 6, 3, "cpu")
    w = dec.lstm.weight_hh_l0                       # (128,32) orthogonal columns (decoder.py:110-112)
    assert torch.allclose(w.t() @ w, torch.eye(32), atol=1e-4)
    assert 0.5 < float(dec.lstm.bias_ih_l0.std()) < 1.5           # N(0,1) biases (decoder.py:114)
    bound = (6.0 / (dec.out_layer.weight.shape[0] + dec.out_layer.weight.shape[1])) ** 0.5
    assert float(dec.out_layer.weight.abs().max()) <= bound + 1e-6    # Xavier-uniform (decoder.py:116)


def test_config_keeps_reference_names_and_json_round_trip(tmp_path):
    from config import Config
    cfg = Config(make_dirs=False)
    assert cfg.context_max_lenth == 283 and cfg.av_max_length == 101 and cfg.question_max_length == 21
    assert cfg.lr == 1e-4 and cfg.audio_emb == 128 and cfg.video_hidden_dim == 512 and cfg.flatten_dim == 1000
    assert cfg.text_lstm_layers == cfg.dec_lstm_layers == 3 and cfg.text_lstm_dropout == 0.2
    assert str(cfg.dec_model_path).endswith("results/test/dec_model.pth")
    old_out = Config.output_path
    try:
        Config.output_path = tmp_path
        cfg.save_config()
        data = json.load(open(tmp_path / "config.json"))
        assert data["context_max_lenth"] == 283 and data["optim"] == "adam" and isinstance(data["vocab_file"], str)
        data["lr"] = 5e-4
        data["vocab_file"] = "elsewhere/vocab.json"
        json.dump(data, open(tmp_path / "c2.json", "w"))
        cfg2 = Config(str(tmp_path / "c2.json"), make_dirs=False)
        assert cfg2.lr == 5e-4 and str(cfg2.vocab_file) == "elsewhere/vocab.json" and cfg2.optim == "adam"
    finally:
        Config.output_path = old_out
        Config.lr = 1e-4
        from pathlib import Path
        Config.vocab_file = Path("data") / "vocab.json"


def test_synthetic_batches_are_seeded_and_reserve_special_ids():
    from mmqg_amd.synthetic import WORKLOADS, Workload, synthetic_batch
    w = Workload("t", batch=6, n_frames=4, frame_dim=16, audio_dim=8, ctx_len=9, tgt_len=5, vocab=40)
    a, b = synthetic_batch(w, seed=3, ragged=True), synthetic_batch(w, seed=3, ragged=True)
    assert all(torch.equal(a[k], b[k]) for k in a)
    for i in range(6):
        n = int(a["tgt_len"][i])
        assert int(a["target"][i, n - 1]) == 2 and (a["target"][i, :n - 1] >= 3).all() and (a["target"][i, n:] == 0).all()
        assert (a["frames"][i, int(a["n_frames"][i]):] == 0).all()
    assert WORKLOADS["config2"].batch == 64 and WORKLOADS["config2"].text_max_length == 283
    assert WORKLOADS["config5"].vocab == 50000 and WORKLOADS["config5"].hidden == 1024


def test_bleu_restatement_known_values():
    from mmqg_amd.metrics import reference_bleu_scores, sentence_bleu, truncate_at_end
    ref = "the cat is on the mat".split()
    assert sentence_bleu([ref], ref) == pytest.approx(1.0)
    assert sentence_bleu([ref], "the cat".split(), (1, 0, 0, 0)) == pytest.approx(2.718281828 ** (1 - 6 / 2))   # brevity penalty only
    hyp = "the the the the the the the".split()
    assert sentence_bleu([ref], hyp, (1, 0, 0, 0)) == pytest.approx(2 / 7)           # clipped unigram precision
    assert sentence_bleu([ref], "dog".split()) == 0.0
    # the reference's call: every reference is ONE WORD iterated as characters (train.py:115)
    s = reference_bleu_scores("what is a", ["a", "b"])
    assert s["bleu_1"] == pytest.approx(0.5) and 0.0 < s["bleu"] < 1e-50
    assert truncate_at_end([5, 7, 2, 9], 2) == [5, 7]


def test_bleu_matches_the_values_nltk_publishes():
    """nltk is absent here, but its docstrings publish exact results (nltk.translate.bleu_score:
    sentence_bleu, corpus_bleu, modified_precision doctests); the restatement must hit them."""
    from mmqg_amd.metrics import sentence_bleu
    hyp1 = "It is a guide to action which ensures that the military always obeys the commands of the party".split()
    hyp2 = "It is to insure the troops forever hearing the activity guidebook that party direct".split()
    ref1 = "It is a guide to action that ensures that the military will forever heed Party commands".split()
    ref2 = ("It is the guiding principle which guarantees the military forces always being under the command of the "
            "Party").split()
    ref3 = "It is the practical guide for the army always to heed the directions of the party".split()
    refs = [ref1, ref2, ref3]
    assert sentence_bleu(refs, hyp1) == pytest.approx(0.5045666840058485, rel=1e-12)        # sentence_bleu doctest
    assert round(sentence_bleu(refs, hyp1, (1. / 5.,) * 5), 4) == 0.3920                      # custom-weights doctest
    # corpus_bleu doctest: the average of the two sentence scores is 0.6223...
    hyp_b = "he read the book because he was interested in world history".split()
    ref_b = "he was interested in world history because he read the book".split()
    assert (sentence_bleu(refs, hyp1) + sentence_bleu([ref_b], hyp_b)) / 2 == pytest.approx(0.6223247442490669, rel=1e-12)
    # modified_precision doctests (unigram 17/18, bigram 10/17 for hyp1; 8/14 and 1/13 for hyp2), through the
    # n-gram weights; hyp1 is as long as its closest reference, so its brevity penalty is 1
    assert sentence_bleu(refs, hyp1, (1, 0, 0, 0)) == pytest.approx(0.9444444444444444, rel=1e-12)
    assert sentence_bleu(refs, hyp1, (0, 1, 0, 0)) == pytest.approx(0.5882352941176471, rel=1e-12)
    import math
    bp2 = math.exp(1 - 16 / 14)                      # closest reference length to 14 words is 16
    assert sentence_bleu(refs, hyp2, (1, 0, 0, 0)) == pytest.approx(bp2 * 0.5714285714285714, rel=1e-12)
    assert sentence_bleu(refs, hyp2, (0, 1, 0, 0)) == pytest.approx(bp2 * 0.07692307692307693, rel=1e-12)
    the7 = "the the the the the the the".split()
    assert sentence_bleu(["the cat is on the mat".split(), "there is a cat on the mat".split()], the7,
                         (1, 0, 0, 0)) == pytest.approx(0.2857142857142857, rel=1e-12)


def test_bench_launches_itself_for_more_than_one_gpu(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE starts the one-node launcher as a child."""
    import bench
    assert bench.self_launch_needed(8, {}) and bench.self_launch_needed(2, {"RANK": "0"})
    assert not bench.self_launch_needed(8, {"WORLD_SIZE": "8"}) and not bench.self_launch_needed(1, {})
    argv = ["--gpus", "8", "--steps", "7", "--warmup", "2"]
    cmd = bench.launcher_command(8, argv, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                              # the ranks see the caller's own flags
    assert 1024 < bench.free_port() < 65536
    # on a box with fewer devices than asked for the parent says so and exits non-zero without spawning ranks
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 64" in r.stderr and "device(s)" in r.stderr


def test_bench_launcher_fails_fast_when_a_rank_dies_or_hangs():
    """`bench.py --gpus N` must not hang or print a half result when a rank is lost: one rank dies before
    it touches the GPU (test hook) -> the launcher tears the others down, the parent prints ONE `FAILED` line naming the
    reason, no JSON line, non-zero exit code; a launch that does not finish within --launch-timeout is killed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MMQG_BENCH_TESTING="1", MMQG_BENCH_FAKE_DEVICES="2", MMQG_BENCH_TEST_RANK_FAIL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    failed = [ln for ln in r.stderr.splitlines() if ln.startswith("bench.py: FAILED")]
    assert len(failed) == 1 and "ranks died" in failed[0], r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no result line may be printed"
    # a launch that exceeds its time limit is killed with its whole process group
    env.pop("MMQG_BENCH_TEST_RANK_FAIL")
    env["MMQG_BENCH_TEST_RANK_HANG"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--launch-timeout", "20"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3
    failed = [ln for ln in r.stderr.splitlines() if ln.startswith("bench.py: FAILED")]
    assert len(failed) == 1 and "did not finish within 20 s" in failed[0], r.stderr[-1500:]
