#!/usr/bin/env python3
"""Fixed vs per-chunk cost of the split-bf16 ping-pong kernel: one round of 256 x 128 tiles at several K (run with
MMQG_X3_MAX_SPLIT=1 so that no k slices are made), for the three operand layouts the step uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import ops


def t(al, bl, M, N, K, beta, iters=30):
    A = torch.randn((M, K) if al == 0 else (K, M), device="cuda")
    B = torch.randn((N, K) if bl == 0 else (K, N), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    f = lambda: ops.gemm(al, bl, M, N, K, A, A.stride(0), B, B.stride(0), C, N, beta=beta)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("env", {k: v for k, v in os.environ.items() if k.startswith("MMQG_")})
for name, al, bl, M, N, beta in (("TN wgrad 2048x2048 (128 tiles), beta=1", 1, 1, 2048, 2048, 1), ("NT 2048x2048, beta=0", 0, 0, 2048, 2048, 0),
                                 ("NN 2048x2048, beta=0", 0, 1, 2048, 2048, 0), ("TN 4096x4096 (512 tiles = 2 rounds), beta=1", 1, 1, 4096, 4096, 1)):
    prev = None
    for K in (512, 1024, 2048, 4096):
        us = t(al, bl, M, N, K, beta)
        per = "" if prev is None else f"   per 32-k chunk since the previous K: {(us - prev[1]) / ((K - prev[0]) / 32):.3f} us"
        print(f"{name:48s} K={K:5d} {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s{per}", flush=True)
        prev = (K, us)
