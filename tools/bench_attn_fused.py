#!/usr/bin/env python3
"""Fused score product + softmax + context launch (csrc/attention_fused.hip) against the unfused attention launch, on
synthetic value tensors in the fused per-question layout.  python tools/bench_attn_fused.py [B] [Lt] [Lav] [H]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import _lib, ops


def run(B, Lt=283, Lav=101, H=512, Da=128, Dv=512, iters=300, T=20, E=300):
    lib = _lib.load()
    stride = Lt * H + Lav * Da + Lav * Dv
    vals = torch.randn(B, stride, device="cuda")
    S = Lt + 2 * Lav
    ldS = (S + 3) // 4 * 4
    Cw = H + Da + Dv
    sc = torch.randn(T, B, ldS, device="cuda")
    at = torch.empty(T, B, ldS, device="cuda")
    cx = torch.empty(T, B, Cw, device="cuda")
    W = torch.randn(S, E + H, device="cuda") * H ** -0.5
    h = torch.randn(T, B, H, device="cuda")
    v = _lib.AttnValues()
    v.B, v.Lt, v.Lav, v.H, v.Da, v.Dv = B, Lt, Lav, H, Da, Dv
    v.text, v.audio, v.video = vals.data_ptr(), vals.data_ptr() + 4 * Lt * H, vals.data_ptr() + 4 * (Lt * H + Lav * Da)
    v.text_stride_b = v.audio_stride_b = v.video_stride_b = stride
    n = int(lib.mmqg_attn_fused_ws_bytes(C.byref(v), H))
    ws = torch.zeros((n + 3) // 4, device="cuda")
    s = ops._stream()

    def plain(i):
        t = i % T
        _lib.check(lib.mmqg_attn_softmax_context_fwd(C.byref(v), sc[t].data_ptr(), ldS, at[t].data_ptr(), ldS, cx[t].data_ptr(), Cw, s))

    def fused(i):
        t = i % T
        rc = lib.mmqg_attn_scores_softmax_context_fwd(C.byref(v), sc[t].data_ptr(), ldS, h[t].data_ptr(), H, W.data_ptr() + 4 * E,
                                                      E + H, H, at[t].data_ptr(), ldS, cx[t].data_ptr(), Cw, ws.data_ptr(), n, s)
        assert rc == 0
    out = []
    for fn in (plain, fused):
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record(); e1.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    nbytes = B * 4 * (stride + 2 * S + Cw)
    print(f"B={B} Lt={Lt} Lav={Lav} H={H}: attention alone {out[0]:.2f} us ({nbytes/out[0]/1e3/8000:.3f} of 8 TB/s); "
          f"fused with the score product {out[1]:.2f} us ({(nbytes + 4 * S * H)/out[1]/1e3/8000:.3f})", flush=True)


if len(sys.argv) > 1:
    run(*[int(x) for x in sys.argv[1:]])
else:
    run(64)
    run(32)
    run(128, H=1024, Dv=1024)
