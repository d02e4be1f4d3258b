#!/usr/bin/env python3
"""Attention forward/backward kernel timing on synthetic value tensors (fused per-question layout)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import _lib, ops

def run(B, Lt, Lav, H=512, Da=128, Dv=512, iters=300, T=20):
    lib = _lib.load()
    stride = Lt * H + Lav * Da + Lav * Dv
    vals = torch.randn(B, stride, device="cuda")
    S = Lt + 2 * Lav
    ldS = (S + 3) // 4 * 4
    Cw = H + Da + Dv
    sc = torch.randn(T, B, ldS, device="cuda")
    at = torch.empty(T, B, ldS, device="cuda")
    cx = torch.empty(T, B, Cw, device="cuda")
    dcx = torch.randn(T, B, Cw, device="cuda")
    ds = torch.zeros(T, B, ldS, device="cuda")
    v = _lib.AttnValues()
    v.B, v.Lt, v.Lav, v.H, v.Da, v.Dv = B, Lt, Lav, H, Da, Dv
    v.text, v.audio, v.video = vals.data_ptr(), vals.data_ptr() + 4 * Lt * H, vals.data_ptr() + 4 * (Lt * H + Lav * Da)
    v.text_stride_b = v.audio_stride_b = v.video_stride_b = stride
    s = ops._stream()
    def fwd(i):
        t = i % T
        _lib.check(lib.mmqg_attn_softmax_context_fwd(C.byref(v), sc[t].data_ptr(), ldS, at[t].data_ptr(), ldS, cx[t].data_ptr(), Cw, s))
    def bwd(i):
        t = i % T
        _lib.check(lib.mmqg_attn_context_bwd(C.byref(v), at[t].data_ptr(), ldS, dcx[t].data_ptr(), Cw, None, 0, ds[t].data_ptr(), ldS, s))
    out = []
    for fn in (fwd, bwd):
        for i in range(10): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fn(i)
        e1.record(); e1.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    nbytes = B * 4 * (stride + 2 * S + Cw)
    print(f"B={B} Lt={Lt} Lav={Lav}: fwd {out[0]:.2f} us = {nbytes/out[0]/1e3:.0f} GB/s ({nbytes/out[0]/1e3/8000:.3f} of 8 TB/s); bwd {out[1]:.2f} us (2 kernels) = {nbytes/out[1]/1e3:.0f} GB/s", flush=True)

print("chunk", os.environ.get("MMQG_ATTN_CHUNK", "128"))
run(64, 283, 101)
run(32, 283, 101)
run(64, 32, 8)
run(128, 283, 101, H=1024, Dv=1024)
run(512, 283, 101)
