#!/usr/bin/env python3
"""Accuracy (against float64) and time of the large-GEMM shapes of a step, for whichever kernel MMQG_GEMM_X3 selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import ops


def run(name, al, bl, M, N, K, beta=0, bias=False, iters=20):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K) if al == 0 else (K, M), device="cuda", generator=g)
    B = torch.randn((N, K) if bl == 0 else (K, N), device="cuda", generator=g)
    C0 = torch.randn(M, N, device="cuda", generator=g)
    bv = torch.randn(N, device="cuda", generator=g) if bias else None
    C = C0.clone()
    ops.gemm(al, bl, M, N, K, A, A.stride(0), B, B.stride(0), C, N, beta=beta, bias=bv)
    Ad = A.double() if al == 0 else A.double().T
    Bd = B.double().T if bl == 0 else B.double()
    ref = Ad @ Bd + (C0.double() if beta else 0) + (bv.double() if bias else 0)
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    f = lambda: ops.gemm(al, bl, M, N, K, A, A.stride(0), B, B.stride(0), C, N, beta=0, bias=bv)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:30s} M={M:5d} N={N:5d} K={K:5d} beta={beta} err/max|ref|={err:.2e}  {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s", flush=True)


print("MMQG_GEMM_X3 =", os.environ.get("MMQG_GEMM_X3", "(default on)"), " MMQG_NO_NT_TILE =", os.environ.get("MMQG_NO_NT_TILE", "0"))
run("vocab fwd (NT)", 0, 0, 1280, 10000, 512, bias=True)
run("vocab fwd config5 (NT)", 0, 0, 2560, 50000, 1024, bias=True, iters=5)
run("vocab dgrad (NN)", 0, 1, 1280, 512, 10000)
run("vocab wgrad (TN)", 1, 1, 10000, 512, 1280, beta=1)
run("dec wgrad W_ih0 (TN)", 1, 1, 2048, 1152, 1280, beta=1)
run("lstm wgrad W_hh (TN)", 1, 1, 2048, 512, 2048, beta=1)
run("text hoist layer0 (NT)", 0, 0, 2048, 2048, 300, bias=True)
run("dec hoist gates (NT)", 0, 0, 1280, 2048, 300, bias=True)
run("frame hoist (NT)", 0, 0, 512, 2048, 2048)
run("ragged (NT)", 0, 0, 1000, 777, 300)
run("ragged (NN)", 0, 1, 333, 485, 2048)
run("ragged (TN)", 1, 1, 485, 300, 1280, beta=1)
run("square 4096 (NT)", 0, 0, 4096, 4096, 4096, iters=5)
run("square 4096 (TN)", 1, 1, 4096, 4096, 4096, iters=5)
