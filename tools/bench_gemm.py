#!/usr/bin/env python3
"""Timing of the large-GEMM shapes of one training step (config 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import ops

def t(name, al, bl, M, N, K, iters=30):
    A = torch.randn(M * K, device="cuda"); B = torch.randn(N * K, device="cuda"); C = torch.zeros(M, N, device="cuda")
    lda = K if al == 0 else M
    ldb = K if bl == 0 else N
    f = lambda: ops.gemm(al, bl, M, N, K, A, lda, B, ldb, C, N, beta=0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:34s} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s", flush=True)

print("BIG_BK", os.environ.get("MMQG_GEMM_BIG_BK", "per layout"))
t("vocab fwd  (NT)", 0, 0, 1280, 10000, 512)
t("vocab fwd config5 (NT)", 0, 0, 2560, 50000, 1024, iters=10)
t("vocab dgrad (NN)", 0, 1, 1280, 512, 10000)
t("vocab wgrad (TN)", 1, 1, 10000, 512, 1280)
t("dec wgrad W_ih0 (TN)", 1, 1, 2048, 1152, 1280)
t("lstm wgrad W_hh (TN)", 1, 1, 2048, 512, 2048)
t("text hoist layer0 (NT)", 0, 0, 2048, 2048, 300)
t("dec hoist gates (NT)", 0, 0, 1280, 2048, 300)
t("frame hoist (NT)", 0, 0, 512, 2048, 2048)
t("square 4096 (NT)", 0, 0, 4096, 4096, 4096, iters=5)
