#!/usr/bin/env python3
"""Benchmark of the hot path: training questions/s (forward + backward + optimizer) of the
multimodal encoder -> attention decoder step on synthetic tensors, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config4|config5|...]

N > 1: either launch it under ``python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or just run
``python bench.py --gpus N``: with no WORLD_SIZE in the environment the script starts that
launcher itself as a child process — before anything here has touched the GPU — and relays rank
0's JSON line and the child's exit code.  Every rank trains its own shard of the global batch
(weak scaling, B questions per GPU) and gradients are all-reduced over RCCL.  Rank 0 prints ONE
JSON line.

Besides throughput the line carries
  roofline      the decoder-attention kernel (softmax + context, the HBM-bound kernel BASELINE.json's
                metric names): algorithmic bytes per launch / average launch duration, measured here
                with HIP events on the launch stream.  Two durations: ``achieved`` = back-to-back
                launches on the step's own buffers (what the step sees: the 54 MB value tensor of a
                batch is re-read every decode step and stays in the 256 MiB Infinity Cache), and
                ``achieved_beyond_mall`` = the same launches rotated over enough distinct value
                tensors (> 2 x 256 MiB) that every row comes from HBM;
                Where the decoder's forward time loop runs as ONE persistent launch (csrc/persist_dec.hip: B <= 64,
                three layers, weights within the chip's LDS) no attention launch exists in the step any more: the
                attention is a phase of that kernel, and ``roofline`` is then that phase — the same algorithmic bytes
                per token over the window from the first value-row load to the last context stored, read from the
                device wall-clock stamps of the kernel's stamped instantiation — with the whole launch's duration
                (HIP events; rocprofv3's average for decoder_persist_fwd_kernel agrees) beside it and the standalone
                kernel's figures under ``standalone_kernel``;
  roofline_mfma the vocabulary-projection GEMM (fp32 MFMA), same method;
  cpu_baseline  the CPU oracle (``oracle/``: the reference's batch-1 loop restated, kind "port")
                timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA (spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md; the 5 PF headline figure is 2:1 sparse)
MALL_BYTES = 256 << 20         # Infinity Cache


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--batch", type=int, default=0, help="questions per GPU (default: the workload's)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget")
    ap.add_argument("--kernel-iters", type=int, default=200, help="launches per kernel-duration measurement")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="seconds the self-launched ranks may take in all")
    ap.add_argument("--skip-zero-rows", action="store_true",
                    help="attention kernels skip the zero-padded value rows (identical results; NOT the default: the "
                         "roofline is defined on streaming the padded extents)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------- self-launch
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(gpus: int, argv, port: int):
    """The one-node launch the driver would use: one rank per GPU, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch_needed(gpus: int, env) -> bool:
    # MMQG_BENCH_FORCE_LAUNCH=1: go through the launcher for one GPU too (rehearses the N > 1 path on a one-GPU box)
    return (gpus > 1 or env.get("MMQG_BENCH_FORCE_LAUNCH") == "1") and "WORLD_SIZE" not in env


def descendants(pid: int):
    """Every live descendant of pid (exact PIDs), from /proc: the launcher's ranks sit in sessions of their own, so a
    process-group kill would miss them."""
    kids = {}
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        try:
            with open(f"/proc/{name}/stat") as f:
                ppid = int(f.read().rsplit(")", 1)[1].split()[1])
        except (OSError, ValueError, IndexError):
            continue
        kids.setdefault(ppid, []).append(int(name))
    out, todo = [], [pid]
    while todo:
        for c in kids.get(todo.pop(), []):
            out.append(c)
            todo.append(c)
    return out


def self_launch(a, argv) -> int:
    """Start the N ranks as a child process tree.  Nothing in this process has touched the GPU yet
    (``import torch`` and ``device_count`` do not), and this process never execs: it waits for the
    launcher and hands its exit code on.  Fails fast with a one-line reason and a non-zero exit code when the node
    has fewer devices than ranks, when any rank dies (the launcher then tears the others down) or when the ranks do
    not finish within --launch-timeout seconds (a rank stuck in rendezvous or in a collective)."""
    testing = os.environ.get("MMQG_BENCH_TESTING") == "1"         # the test hooks below work only with this set
    have = (int(os.environ.get("MMQG_BENCH_FAKE_DEVICES", "0")) if testing else 0) or torch.cuda.device_count()
    if have < a.gpus:
        print(f"bench.py: FAILED: --gpus {a.gpus} but this node shows {have} device(s)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(a.gpus, argv, free_port())
    proc = subprocess.Popen(cmd, env=env)
    try:
        rc = proc.wait(timeout=a.launch_timeout)
    except subprocess.TimeoutExpired:
        import signal
        # the launcher puts every rank into a session of its own: collect the whole tree (exact PIDs) before killing
        victims = [proc.pid] + descendants(proc.pid)
        for pid in reversed(victims):
            try:
                os.kill(pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        proc.wait()
        left = [p for p in victims if os.path.exists(f"/proc/{p}")]
        if left:
            print(f"bench.py: processes still alive after SIGKILL: {left}", file=sys.stderr)
        print(f"bench.py: FAILED: the {a.gpus} ranks did not finish within {a.launch_timeout:.0f} s (a rank stuck in rendezvous "
              f"or in a collective); the launcher and its ranks were killed", file=sys.stderr)
        return 3
    if rc != 0:
        print(f"bench.py: FAILED: the launcher exited with code {rc}: at least one of the {a.gpus} ranks died (its traceback is "
              f"above); no result line was printed", file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    return 0


# --------------------------------------------------------------------------------- rooflines
def attention_bytes(w, B):
    """Algorithmic bytes of ONE attention launch (forward, one decode step, B questions), SURVEY §8d:
    value rows + query-side i/o; the score matrix is counted by the score GEMM, not here."""
    vals = 4 * (w.text_max_length * w.hidden + w.av_max_length * w.audio_dim + w.av_max_length * w.video_hidden)
    S = w.text_max_length + 2 * w.av_max_length
    io = 4 * (2 * S + (w.hidden + w.audio_dim + w.video_hidden))      # scores in, weights out, context out
    return B * (vals + io), vals


def time_launches(fn, iters):
    """Average duration of fn() launches enqueued back to back, by HIP events on the current
    stream (the stream the kernels are launched on)."""
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3      # seconds


def source_sha(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def recorded_traffic(workload_key, source="attention.hip"):
    """PMC-measured HBM-side bytes per launch (profiles/attn_traffic.json, written by tools/pmc_traffic.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes).  Only a record taken on the kernel source that is
    loaded now counts; anything else reports null."""
    tpath = os.path.join(ROOT, "profiles", "attn_traffic.json")
    try:
        rec = json.load(open(tpath)).get(workload_key)
    except Exception:
        return None, "no profiles/attn_traffic.json"
    if not rec:
        return None, f"no PMC record for {workload_key}"
    if rec.get("source_sha") != source_sha("multi-modal-qg_amd/csrc/" + source):
        return None, "PMC record was taken on a different " + source
    return rec.get("hbm_bytes_per_launch"), f"{rec.get('kernel')} grid {rec.get('grid')}"


def recorded_in_step(workload_key, source="attention.hip"):
    """In-step duration of the attention forward launch — or, where the decoder's forward loop is one persistent
    launch, of that launch — (profiles/attn_in_step.json, written by tools/attn_in_step.py from a rocprofv3
    kernel-trace of `bench.py --kernel-iters 0`): only a record taken on the kernel source that is loaded now counts."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "attn_in_step.json"))).get(workload_key)
    except Exception:
        return None
    if not rec or rec.get("source_sha") != source_sha("multi-modal-qg_amd/csrc/" + source):
        return None
    return rec


def decoder_persist_probe(tr, reps=3):
    """The decoder's persistent forward launch under its stamped instantiation (mmqg_decoder_persist_set_trace): per
    (workgroup, token) device wall-clock stamps at 100 MHz.  Returns None when the step does not run that kernel."""
    from mmqg_amd import _lib, ops
    lib = _lib.load()
    if not tr.d_dec.persist_ws:
        return None
    G, T = 256, tr.Td
    buf = torch.zeros(G * T * 8, device=tr.dev, dtype=torch.int64)
    n0 = lib.mmqg_decoder_persist_launch_count()
    tr.d_dec.phase = 2
    try:
        for _ in range(reps):
            _lib.check(lib.mmqg_decoder_persist_set_trace(buf.data_ptr(), buf.numel()))
            _lib.check(lib.mmqg_decoder_seq_fwd(C.byref(tr.d_dec), ops._stream()))
            _lib.check(lib.mmqg_decoder_persist_set_trace(None, 0))
            torch.cuda.synchronize()
    finally:
        tr.d_dec.phase = 0
    if lib.mmqg_decoder_persist_launch_count() != n0 + reps:
        return None
    t = buf.view(G, T, 8).cpu().double() * 0.01            # us
    t = t[t[:, 0, 0] > 0]                                  # workgroups that ran (min(CUs, 256))
    mid = slice(2, T - 2) if T > 6 else slice(0, T)
    start = t[:, :, 0].min(0).values
    period = float((start[1:] - start[:-1])[mid].mean()) if T > 1 else float("nan")
    # attention window of a token: from the first workgroup that has arrived at the score barrier (its waves then issue
    # the first loads of their value rows) to the last workgroup's contexts stored
    win = (t[:, :, 3].max(0).values - t[:, :, 1].min(0).values)[mid]
    after = (t[:, :, 3].max(0).values - t[:, :, 2].min(0).values)[mid]
    return {"workgroups": int(t.shape[0]), "tokens": T, "us_per_token": round(period, 2),
            "attention_window_us": round(float(win.mean()), 2), "attention_after_scores_us": round(float(after.mean()), 2),
            "loop_us": round(float(t[:, -1, 7].max() - t[:, 0, 0].min()), 1)}


def graph_time(fn, reps=10):
    """Average GPU time of fn() captured into a hipGraph and replayed (what the step does), HIP events."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def loop_rooflines(tr, w):
    """The four dependent time loops of the step (75% of its time), each replayed alone as a graph: time per token /
    per wavefront diagonal, the arithmetic and the distinct operand bytes of that unit, and what fraction of the fp32
    MFMA peak / of the HBM peak that is.  Neither bound is near: these loops are bound by the number of dependent
    stages (launch boundaries or device-wide barriers) per unit."""
    from mmqg_amd import _lib, ops
    lib = _lib.load()
    B, H, L, E = tr.B, tr.H, tr.L, tr.E
    Tc, Td, S, Cw = tr.Tc, tr.Td, tr.S, tr.Cw
    vals = 4 * (tr.Lt * H + tr.Lav * tr.Da + tr.Lav * tr.Dv)

    def s():
        return ops._stream()

    def dec_fwd():
        tr.d_dec.phase = 2
        _lib.check(lib.mmqg_decoder_seq_fwd(C.byref(tr.d_dec), s()))
        tr.d_dec.phase = 0

    def dec_bwd():
        tr.g_dec.phase = 1
        _lib.check(lib.mmqg_decoder_seq_bwd(C.byref(tr.d_dec), C.byref(tr.g_dec), s()))
        tr.g_dec.phase = 0

    def text_fwd():
        _lib.check(lib.mmqg_lstm_seq_fwd(C.byref(tr.d_text), s()))

    def text_bwd():
        tr.g_text.phase = 1
        _lib.check(lib.mmqg_lstm_seq_bwd(C.byref(tr.d_text), C.byref(tr.g_text), s()))
        tr.g_text.phase = 0

    lstm_w = 4 * (4 * H * (Cw + H) + (L - 1) * 4 * H * 2 * H)            # decoder recurrent weights, bytes
    dec_flop = 2.0 * B * (S * H + 4 * H * (Cw + H) + (L - 1) * 4 * H * 2 * H) + 2.0 * B * vals / 4
    dec_bytes = lstm_w + 4 * S * H + B * vals
    text_flop = 2.0 * B * (4 * H * H + (L - 1) * 4 * H * 2 * H)           # per diagonal, all layers (input product hoisted)
    out = {}
    for name, fn, units, unit, flop, nbytes, stages in (
            ("decoder_fwd", dec_fwd, Td, "token", dec_flop, dec_bytes,
             "persistent: 5 device-wide barriers per token, recurrent weights resident in LDS" if tr.d_dec.persist_ws
             else "5 dependent launches per token"),
            ("decoder_bwd", dec_bwd, Td, "token", 2 * dec_flop, dec_bytes + B * vals,
             "persistent: 5 device-wide barriers per token (score-gradient product, three cell stages whose k-slices the "
             "consumer sums, attention backward), recurrent weights resident in LDS; includes the two value-gradient kernels "
             "behind the loop" if tr.g_dec.persist_ws else "5 dependent launches per token"),
            ("text_encoder_fwd", text_fwd, Tc + L - 1, "diagonal", text_flop, 4 * B * 2 * H * L,
             "persistent: 1 device-wide barrier per diagonal, weights resident in LDS"),
            ("text_encoder_bwd", text_bwd, Tc + L - 1, "diagonal", text_flop, 4 * B * 8 * H * L,
             "persistent: 2 device-wide barriers per diagonal, weights resident in LDS")):
        dt = graph_time(fn)
        per = dt / units
        out[name] = {"us_per_" + unit: round(per * 1e6, 2), "ms": round(dt * 1e3, 4), "gflop_per_" + unit: round(flop / 1e9, 3),
                     "achieved_tflops": round(flop / per / 1e12, 2),
                     "frac_of_fp32_mfma_peak": round(flop / per / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                     "distinct_operand_bytes_per_" + unit: int(nbytes), "achieved_gbs": round(nbytes / per / 1e9, 1),
                     "frac_of_hbm_peak": round(nbytes / per / 1e9 / HBM_PEAK_GBS, 4), "bound": "latency: " + stages}
    out["persistent_launches"] = {"forward": int(lib.mmqg_persist_launch_count()), "backward": int(lib.mmqg_persist_bwd_launch_count()),
                                  "decoder_forward": int(lib.mmqg_decoder_persist_launch_count()),
                                  "decoder_backward": int(lib.mmqg_decoder_persist_bwd_launch_count())}
    return out


def kernel_rooflines(tr, w, iters):
    from mmqg_amd import _lib, ops
    lib, s = _lib.load(), ops._stream()
    d = tr.d_dec
    ws = tr.ws
    B, Td, ldS, Cw = tr.B, tr.Td, tr.ldS, tr.Cw
    sc, at, cx = ws["scores"], ws["attn"], ws["ctx"]

    # the roofline is defined on the kernel that streams every value row (the padded extents the reference's bmm
    # reads): always measured with zero_past_len = 0, whatever the step itself was run with
    vstream = type(d.values)()
    C.memmove(C.addressof(vstream), C.addressof(d.values), C.sizeof(vstream))
    vstream.zero_past_len = 0

    def attn(i):
        t = i % Td
        _lib.check(lib.mmqg_attn_softmax_context_fwd(C.byref(vstream), sc[t].data_ptr(), ldS, at[t].data_ptr(), ldS,
                                                     cx[t].data_ptr(), Cw, s))
    dt = time_launches(attn, iters)
    nbytes, _ = attention_bytes(w, B)
    key = w.name.split(":")[0].split(" ")[0]
    traffic, note = recorded_traffic(key)

    # the same launch with every value row coming from HBM: rotate over distinct value tensors whose total
    # is more than twice the Infinity Cache, so a tensor is long evicted when its turn comes again
    vbytes = ws["values"].numel() * 4
    n_rot = max(3, -(-2 * MALL_BYTES // vbytes) + 1)
    rot = torch.randn(n_rot, ws["values"].numel(), device=ws["values"].device)
    descs = []
    for r in range(n_rot):
        v = type(d.values)()
        C.memmove(C.addressof(v), C.addressof(vstream), C.sizeof(v))
        base = rot[r].data_ptr()
        v.text, v.audio, v.video = base, base + 4 * tr.off_audio, base + 4 * tr.off_video
        descs.append(v)

    def attn_cold(i):
        t = i % Td
        _lib.check(lib.mmqg_attn_softmax_context_fwd(C.byref(descs[i % n_rot]), sc[t].data_ptr(), ldS, at[t].data_ptr(),
                                                     ldS, cx[t].data_ptr(), Cw, s))
    dtc = time_launches(attn_cold, max(iters, 4 * n_rot))
    del rot
    # what the step itself runs: the launches between the score product and the layer-0 cell of every decode step (cold
    # L2: the layer-step kernels in between stream the LSTM weights through it), from the rocprofv3 kernel-trace of
    # `bench.py --kernel-iters 0` recorded under profiles/ (hash-checked against the loaded attention.hip)
    rec = recorded_in_step(key)
    dt_b2b = dt
    basis = "back-to-back launches on the step's own buffers, HIP events (no in-step record for this attention.hip)"
    if rec:
        dt = rec["us_per_launch"] * 1e-6
        basis = ("in-step launches: rocprofv3 kernel-trace of `bench.py --kernel-iters 0`, %d launches, recorded in "
                 "profiles/attn_in_step.json" % rec["launches"])
    roof = {"kernel": "attn_softmax_context_fwd_kernel<64>", "bound": "hbm", "achieved": round(nbytes / dt / 1e9, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / dt / 1e9 / HBM_PEAK_GBS, 4), "frac_basis": basis,
            "traffic": traffic, "traffic_source": note, "bytes_per_launch": nbytes, "us_per_launch": round(dt * 1e6, 2),
            "us_per_launch_back_to_back": round(dt_b2b * 1e6, 2),
            "achieved_back_to_back": round(nbytes / dt_b2b / 1e9, 1),
            "frac_back_to_back": round(nbytes / dt_b2b / 1e9 / HBM_PEAK_GBS, 4),
            "served_from": "the step's own value tensor, re-read every decode step (Infinity Cache resident "
                           "when it is < 256 MiB)",
            "achieved_beyond_mall": round(nbytes / dtc / 1e9, 1),
            "frac_beyond_mall": round(nbytes / dtc / 1e9 / HBM_PEAK_GBS, 4),
            "us_per_launch_beyond_mall": round(dtc * 1e6, 2),
            "beyond_mall_working_set_bytes": int(n_rot * vbytes)}
    probe = decoder_persist_probe(tr)
    if probe:
        # the step runs no attention launch: the forward attention is a phase of the persistent decoder kernel
        def dec_loop():      # (the stream is looked up inside: a graph capture runs on its own stream)
            d.phase = 2
            _lib.check(lib.mmqg_decoder_seq_fwd(C.byref(d), ops._stream()))
            d.phase = 0
        dtk = graph_time(dec_loop)
        win = probe["attention_window_us"] * 1e-6
        ktraffic, knote = recorded_traffic(key + ":decoder_persist_fwd", "persist_dec.hip")
        krec = recorded_in_step(key, "persist_dec.hip")
        lstm_w = 4 * (4 * tr.H * (Cw + tr.H) + (tr.L - 1) * 4 * tr.H * 2 * tr.H) + 4 * tr.S * tr.H
        kbytes = Td * nbytes + lstm_w + 4 * Td * B * (3 * 4 * tr.H + 3 * 2 * tr.H)      # + weights once + saved gates, h, c
        standalone = roof
        roof = {"kernel": "decoder_persist_fwd_kernel: attention phase (three softmaxes + three contexts of one token, %d "
                          "questions)" % B, "bound": "hbm", "achieved": round(nbytes / win / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(nbytes / win / 1e9 / HBM_PEAK_GBS, 4),
                "frac_basis": "algorithmic bytes of one token's attention / the phase's window inside the persistent launch "
                              "(first workgroup past its score tile, whose waves then issue the first value-row loads, to the "
                              "last workgroup's contexts stored), device wall-clock stamps of the stamped instantiation, "
                              "%d workgroups, mean over the steady-state tokens" % probe["workgroups"],
                "bytes_per_launch": nbytes, "us_per_launch": probe["attention_window_us"],
                "unit_of_launch": "one token's attention phase (the kernel launch holds %d of them)" % Td,
                "window_after_scores_us": probe["attention_after_scores_us"],
                "traffic": None if ktraffic is None else int(ktraffic / Td),
                "traffic_source": knote + ("; per launch / %d tokens (includes the 31 MB of recurrent weights read once and "
                                           "the saved activations)" % Td if ktraffic is not None else ""),
                "whole_launch": {"kernel": "decoder_persist_fwd_kernel", "us_per_launch": round(dtk * 1e6, 1),
                                 "basis": "HIP events over replays of the launch (with its 4 KB barrier-block fill) as a graph",
                                 "us_per_launch_in_step_rocprof": krec["us_per_launch"] if krec else None,
                                 "tokens_per_launch": Td, "us_per_token_stamps": probe["us_per_token"],
                                 "algorithmic_bytes_per_launch": int(kbytes),
                                 "achieved": round(kbytes / dtk / 1e9, 1),
                                 "frac": round(kbytes / dtk / 1e9 / HBM_PEAK_GBS, 4),
                                 "bound": "latency: 5 device-wide barriers per token; only the attention phase streams HBM"},
                "standalone_kernel": standalone}
    # vocabulary projection: logits[Td*B, V] = h_top * W_out^T + b
    R, H, V = Td * B, tr.H, tr.V
    htop = ws["hs_d"][tr.L - 1, 1:].reshape(R, H)
    out = tr.dec.out_layer
    scratch = torch.empty(R, V, device=htop.device)

    tiles = C.c_int32(0)

    def proj(i):     # the launch the step makes: the product with the loss's row statistics in its epilogue
        _lib.check(lib.mmqg_projection_fwd(R, V, H, htop.data_ptr(), H, out.weight.data_ptr(), H, out.bias.data_ptr(),
                                           scratch.data_ptr(), V, tr.ws["proj_stats"].data_ptr(), tr._proj_stats_bytes,
                                           C.byref(tiles), ops._stream()), "projection_fwd")
    dtp = time_launches(proj, max(10, iters // 10))
    flops = 2.0 * R * H * V
    which = int(lib.mmqg_projection_last_kernel())
    tf = flops / dtp / 1e12
    if which == 2:
        # fp32-exact operands as three bf16 pieces each, six bf16 MFMAs per fp32 product: the matrix-core ceiling of
        # the method is the dense bf16 peak / 6; the fraction of the fp32 MFMA peak is given beside it
        peak = MFMA_BF16_PEAK_TFLOPS / 6.0
        mfma = {"kernel": "gemm_x3pp_kernel<NT, stats> (vocab projection fwd: split-bf16, 256 x 128 tiles, loss "
                          "statistics in the epilogue)", "bound": "mfma", "achieved": round(tf, 2),
                "peak": round(peak, 1), "unit": "TFLOP/s (fp32-equivalent: 2 M N K per launch)", "frac": round(tf / peak, 4),
                "peak_basis": "dense bf16 MFMA 2500 TFLOP/s / 6 piece products per fp32 product",
                "executed_bf16_tflops": round(6 * tf, 1), "frac_of_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
    else:
        mfma = {"kernel": "gemm_nt_tile_kernel (vocab projection fwd: one 256 x BN tile per CU, loss statistics in the "
                          "epilogue)" if which == 1 else "gemm_f32 (vocab projection fwd, generic tiles)", "bound": "mfma",
                "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
    mfma.update({"traffic": None, "flops_per_launch": flops, "us_per_launch": round(dtp * 1e6, 2)})
    return roof, mfma


def allreduce_times(tr, iters=5):
    """Per-bucket all-reduce time of the gradient exchange (every rank calls this; RCCL over xGMI for N > 1): each
    bucket of flat_g reduced alone, HIP events on the current stream, average of a few repetitions after one warm-up.
    Not part of the timed step (there the first two buckets travel beside the text encoder's backward)."""
    import torch.distributed as dist
    out = {}
    for name, (lo, hi) in tr.reducer.buckets.items():
        buf = tr.flat_g[lo:hi]
        dist.all_reduce(buf, group=tr.pg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf, group=tr.pg)
        e1.record()
        e1.synchronize()
        out[name] = {"ms": round(e0.elapsed_time(e1) / iters, 4), "bytes": int((hi - lo) * 4)}
    tr.flat_g.zero_()
    return out


# ------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(w, budget_s):
    """The reference's semantics on the host cores: batch-1 loop of zero_grad -> encoders ->
    per-token decoder with teacher forcing -> summed CE -> backward -> Adam (train.py:149-181),
    restated by the oracle.  Bounded: one warm-up question, a thread-count sweep of three questions per
    setting (median decides), then questions until the budget."""
    from mmqg_amd.synthetic import build_models, synthetic_batch
    from oracle import mmqg_oracle as O
    vid, text, dec = build_models(w, "cpu", seed=0)
    sd = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (dec, text, vid)]
    sd[1]["word_embeddings.weight"] = sd[0]["emb_layer.weight"]
    cfg = dict(num_layers=w.layers, hidden_dim=w.hidden, text_max_length=w.text_max_length,
               av_max_length=w.av_max_length, video_hidden_dim=w.video_hidden, start_id=1, end_id=2, mask_mode=0)
    ot = O.OracleTrainer(sd[0], sd[1], sd[2], cfg)
    g = torch.Generator().manual_seed(0)
    H, L, p = w.hidden, w.layers, w.dropout

    def masks(T):
        keep = 1.0 - p
        return [[(torch.rand(1, H, generator=g) < keep).float() / keep for _ in range(L - 1)] for _ in range(T)]

    def one(i):
        b = synthetic_batch(w, seed=100 + i, batch=1)
        b = {k: (v.long() if v.dtype == torch.int32 else v) for k, v in b.items()}
        drop = dict(text=masks(w.ctx_len), dec=masks(w.tgt_len)) if p > 0 else None
        ot.step(b, training=True, drop=drop)

    one(0)
    ncpu = os.cpu_count() or 8
    sweep, best_t, best = {}, 1, None
    t_sweep0 = time.perf_counter()
    for nt in (1, 8, 16):
        if nt > ncpu:
            continue
        torch.set_num_threads(nt)
        times = []
        for r in range(3):
            t0 = time.perf_counter()
            one(1 + r)
            times.append(time.perf_counter() - t0)
        med = sorted(times)[1]
        sweep[str(nt)] = round(1.0 / med, 3)
        if best is None or med < best:
            best, best_t = med, nt
        if time.perf_counter() - t_sweep0 > budget_s:      # a very large workload: stop sweeping
            break
    torch.set_num_threads(best_t)
    n, t0 = 0, time.perf_counter()
    while True:
        one(n + 4)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    return {"value": round(n / el, 4), "unit": "questions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} questions, batch 1, {w.name.split(':')[0]} shapes, fwd+bwd+3xAdam, torch-CPU oracle, "
                      f"{el:.1f} s after warm-up; thread count = best median of 3 questions each at 1/8/16 threads",
            "host_cpus": ncpu, "sweep_qps_by_threads": sweep}


# -------------------------------------------------------------------------------------- main
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if self_launch_needed(a.gpus, os.environ):
        sys.exit(self_launch(a, argv))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner on
    # rank 0's stdout when the first communicator comes up), so everything before the result goes to stderr: file
    # descriptor 1 points at stderr until the line is printed.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import mmqg_amd  # noqa: F401
    from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
    from mmqg_amd.trainer import BatchedTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"bench.py: FAILED: --gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("MMQG_BENCH_TESTING") == "1":                    # test hooks (tests/test_host_cpu.py): off in product runs
        if os.environ.get("MMQG_BENCH_TEST_RANK_FAIL") == str(rank):      # this rank dies before it touches the GPU
            raise SystemExit(17)
        if os.environ.get("MMQG_BENCH_TEST_RANK_HANG") == "1":            # ranks that never finish
            time.sleep(3600)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or (os.environ.get("MMQG_FORCE_DP") == "1" and "RANK" in os.environ)
    backend = None
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from mmqg_amd.distributed import configure_rccl_env
        configure_rccl_env()           # RCCL's channel count fits the CUs the persistent backward leaves free
        torch.distributed.init_process_group("nccl", device_id=dev)
        backend = f"{torch.distributed.get_backend()} (RCCL)"
        if torch.distributed.get_world_size() != world:
            raise SystemExit(f"bench.py: FAILED: the process group has {torch.distributed.get_world_size()} ranks, "
                             f"WORLD_SIZE says {world}")
    w = WORKLOADS[a.workload]
    B = a.batch or w.batch
    vid, text, dec = build_models(w, dev, seed=0)          # same seed on every rank: identical replicas
    tr = BatchedTrainer(vid, text, dec, batch_size=B, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len,
                        lr=1e-4, seed=1234, use_graph=not a.no_graph, skip_zero_value_rows=a.skip_zero_rows).train()
    batches = [synthetic_batch(w, seed=rank * 1000 + i, batch=B) for i in range(4)]    # each rank its own shard
    batches = [{k: v.to(dev) for k, v in b.items()} for b in batches]

    def sync():
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        tr.step(batches[i % len(batches)])
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = tr.step(batches[i % len(batches)])
    sync()
    dt = time.perf_counter() - t0
    loss_val = float(loss)
    if use_pg:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    if not (loss_val == loss_val):
        raise SystemExit("loss is NaN")
    tr.check_health(sync=True)              # a persistent launch of the LAST step that timed out at its barrier raises here
    from mmqg_amd import _lib as _mlib
    _l = _mlib.load()
    health = {"persistent_launch_failures": int(_l.mmqg_persist_failures()),
              "persistent_launches_declined": int(_l.mmqg_persist_declined_count())}

    out = {"metric": "training questions/sec (fwd+bwd+optimizer)", "value": round(world * B * a.steps / dt, 2),
           "unit": "questions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": w.name, "global_batch": world * B, "batch_per_gpu": B, "n_frames": w.n_frames,
                      "frame_dim": w.frame_dim, "audio_dim": w.audio_dim, "ctx_len": w.ctx_len, "tgt_len": w.tgt_len,
                      "vocab": w.vocab, "emb_dim": w.emb_dim, "hidden": w.hidden, "layers": w.layers,
                      "attn_widths": [w.text_max_length, w.av_max_length], "dropout": w.dropout,
                      "parallelism": f"dp{world}", "hipgraph": bool(tr.use_graph),
                      "large_gemms": ("fp32 operands split exactly into 3 bf16 pieces, 6 bf16 MFMAs per product, fp32 accumulation"
                                      if os.environ.get("MMQG_GEMM_X3", "1") != "0" else "fp32 MFMA"),
                      "skip_zero_value_rows": bool(a.skip_zero_rows),
                      "world_size": torch.distributed.get_world_size() if use_pg else 1,
                      "collective_backend": backend},
           "final_loss": round(loss_val, 4), "health": health}
    if use_pg:
        out["config"]["allreduce_ms"] = allreduce_times(tr)
    if rank == 0:
        if a.kernel_iters > 0:
            roof, mfma = kernel_rooflines(tr, w, a.kernel_iters)
            out["roofline"], out["roofline_mfma"] = roof, mfma
            out["roofline_loops"] = loop_rooflines(tr, w)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, a.cpu_seconds)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if use_pg:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    os.dup2(result_fd, 1)
    os.close(result_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
