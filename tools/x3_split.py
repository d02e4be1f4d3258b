#!/usr/bin/env python3
"""Time of the k-sliced shapes of a step for the MMQG_X3_MAX_SPLIT in the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import ops


def run(name, al, bl, M, N, K, beta=0, iters=30):
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
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"split<={os.environ.get('MMQG_X3_MAX_SPLIT','16'):>2s} {name:28s} {us:8.1f} us {2*M*N*K/us/1e6:6.1f} TFLOP/s", flush=True)


run("vocab dgrad (NN)", 0, 1, 1280, 512, 10000)
run("vocab wgrad (TN)", 1, 1, 10000, 512, 1280, beta=1)
run("lstm wgrad W_hh (TN)", 1, 1, 2048, 512, 2048, beta=1)
run("frame hoist (NT)", 0, 0, 512, 2048, 2048)
run("text dx (NN)", 0, 1, 2048, 512, 2048)
run("dgrad config5 (NN)", 0, 1, 2560, 1024, 50000, iters=5)
