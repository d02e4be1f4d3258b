#!/usr/bin/env python3
"""How much does a latency-bound time loop slow down when a GEMM of another stream runs beside it?  (Feasibility check for
pipelining the vocabulary projection / its data gradient by time chunks beside the decoder's loops.)
    python tools/overlap_chain_gemm.py [workload]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import _lib, ops
from mmqg_amd._lib import check
from mmqg_amd.synthetic import WORKLOADS, build_models, synthetic_batch
from mmqg_amd.trainer import BatchedTrainer

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "config2"]
vid, text, dec = build_models(w, "cuda", seed=0)
tr = BatchedTrainer(vid, text, dec, batch_size=w.batch, n_frames=w.n_frames, ctx_len=w.ctx_len, tgt_len=w.tgt_len).train()
b = {k: v.cuda() for k, v in synthetic_batch(w, seed=0).items()}
tr.step(b); tr.forward_backward(b); torch.cuda.synchronize()
lib = _lib.load(); ws = tr.ws
L, B, H, V, Td = tr.L, tr.B, tr.H, tr.V, tr.Td
side = torch.cuda.Stream()

def dec_fwd():
    tr.d_dec.phase = 2
    check(lib.mmqg_decoder_seq_fwd(C.byref(tr.d_dec), ops._stream()))
    tr.d_dec.phase = 0

def dec_bwd():
    tr.g_dec.phase = 1
    check(lib.mmqg_decoder_seq_bwd(C.byref(tr.d_dec), C.byref(tr.g_dec), ops._stream()))
    tr.g_dec.phase = 0

def proj(rows, reps):
    htop = ws["hs_d"][L - 1, 1:].reshape(Td * B, H)
    out = tr.dec.out_layer
    tiles = C.c_int32(0)
    for _ in range(reps):
        check(lib.mmqg_projection_fwd(rows, V, H, htop.data_ptr(), H, out.weight.data_ptr(), H, out.bias.data_ptr(),
                                      ws["logits"].data_ptr(), V, ws["proj_stats"].data_ptr(), tr._proj_stats_bytes, C.byref(tiles), ops._stream()))

def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for name, chain in (("decoder fwd loop", dec_fwd), ("decoder bwd loop", dec_bwd)):
    alone = timed(chain)
    for rows, reps in ((Td * B, 1), (Td * B // 4, 3), (Td * B // 4, 1)):
        g_alone = timed(lambda: proj(rows, reps))
        def both():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                proj(rows, reps)
            chain()
            torch.cuda.current_stream().wait_stream(side)
        t = timed(both)
        print(f"{name}: alone {alone:.0f} us; projection of {rows} rows x{reps} alone {g_alone:.0f} us; both {t:.0f} us "
              f"(serial would be {alone + g_alone:.0f})", flush=True)
