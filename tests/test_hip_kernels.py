"""Kernel-level parity: every HIP kernel, called through the C ABI, against the CPU oracle /
a float64 restatement on the same seeded inputs.  fp32 tolerance 1e-4 (relative to the
magnitude of the expected tensor), index outputs bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def mm():
    import mmqg_amd  # noqa: F401
    from mmqg_amd import _lib, ops
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test on a machine without a ROCm device")
    _lib.load()
    return _lib, ops


from golden_util import close  # noqa: E402  (relative to max|want|, floor 1e-7, logged)


def dev(t):
    return t.cuda()


# ------------------------------------------------------------------------------------ GEMM
GEMM_CASES = [
    # M, N, K, a_layout, b_layout, lda_pad, ldb_pad, ldc_pad, bias, beta, split, K2
    (64, 2048, 512, 0, 0, 0, 0, 0, True, 0, 1, 0),
    (64, 2048, 512, 0, 0, 0, 0, 0, True, 1, -1, 512),      # LSTM gates: dual operand + split-K atomics
    (64, 485, 512, 0, 0, 0, 300, 3, False, 1, -1, 0),      # attention scores: weight sub-block, ldc 488
    (1280, 1000, 512, 0, 0, 0, 0, 0, True, 0, 1, 0),       # vocabulary projection shape (V scaled down)
    (1280, 512, 1000, 0, 1, 0, 0, 0, False, 0, -1, 0),     # dgrad: dlogits * W
    (1000, 512, 1280, 1, 1, 0, 0, 0, False, 1, -1, 0),     # wgrad: dlogits^T * h
    (2048, 1452, 640, 1, 1, 0, 0, 0, False, 1, 1, 0),      # LSTM wgrad, big tiles
    (64, 1152, 2048, 0, 1, 0, 300, 0, False, 0, -1, 0),    # dctx = dgates * W_ih0[:, E:]
    (485, 300, 640, 1, 1, 3, 0, 512, False, 1, -1, 0),     # dW_attn[:, :E] += dS^T * xemb (lda 488, ldc 812)
    (3, 7, 5, 0, 0, 0, 0, 0, True, 0, 1, 0),               # tiny, nothing aligned
    (33, 65, 37, 0, 1, 1, 1, 1, True, 1, 2, 0),            # odd everything, scalar loads
    (65, 33, 37, 1, 0, 1, 1, 1, False, 0, 1, 19),          # m-major A with k-major B, dual
    (130, 257, 300, 0, 0, 0, 0, 0, True, 0, 3, 0),
    (1, 10000, 512, 0, 0, 0, 0, 0, True, 0, -1, 0),        # batch-1 projection (the reference's own shape)
    # shapes the one-tile-per-CU NT kernel takes (csrc/gemm_nt_tile.hip): the full vocabulary projection, ragged edges
    # in both extents with padded leading dimensions, a short K, and a many-tiles-per-CU case
    (1280, 10000, 512, 0, 0, 0, 0, 0, True, 0, -1, 0),
    (1100, 7777, 96, 0, 0, 4, 8, 3, True, 0, 1, 0),
    (2560, 12000, 128, 0, 0, 0, 0, 0, False, 0, -1, 0),
]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_against_float64(mm, case):
    _lib, ops = mm
    M, N, K, al, bl, pa, pb, pc, use_bias, beta, split, K2 = case
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)

    def operand(rows, cols, k_major, pad):
        # k_major: stored [rows][K]; else stored [K][rows]
        shape = (rows, cols + pad) if k_major else (cols, rows + pad)
        full = torch.randn(*shape, generator=g)
        logical = full[:, :cols] if k_major else full[:, :rows].t()
        return full, logical, shape[1]

    Afull, A, lda = operand(M, K, al == 0, pa)
    Bfull, Bt, ldb = operand(N, K, bl == 0, pb)          # Bt logical [N,K]
    want = A.double() @ Bt.double().t()
    A2d = B2d = None
    lda2 = ldb2 = 0
    if K2:
        A2full, A2, lda2 = operand(M, K2, al == 0, 0)
        B2full, B2t, ldb2 = operand(N, K2, bl == 0, 0)
        want = want + A2.double() @ B2t.double().t()
        A2d, B2d = dev(A2full), dev(B2full)
    bias = torch.randn(N, generator=g) if use_bias else None
    bias2 = torch.randn(N, generator=g) if (use_bias and K2) else None
    ldc = N + pc
    Cfull = torch.randn(M, ldc, generator=g)
    if bias is not None:
        want = want + bias.double()
    if bias2 is not None:
        want = want + bias2.double()
    if beta:
        want = want + Cfull[:, :N].double()
    Ad, Bd, Cd = dev(Afull), dev(Bfull), dev(Cfull)
    ops.gemm(al, bl, M, N, K, Ad, lda, Bd, ldb, Cd, ldc, beta=beta, bias=None if bias is None else dev(bias),
             bias2=None if bias2 is None else dev(bias2), A2=A2d, lda2=lda2, B2=B2d, ldb2=ldb2, K2=K2, split_k=split)
    torch.cuda.synchronize()
    got = Cd.cpu()
    close(got[:, :N], want.float(), what=f"gemm{case}")
    if pc:
        assert torch.equal(got[:, N:], Cfull[:, N:]), "gemm wrote outside its N columns"


@pytest.mark.parametrize("al,bl,M,N,K", [(0, 0, 640, 512, 1024), (0, 1, 600, 384, 1000), (1, 1, 777, 300, 1280),
                                          (0, 0, 1280, 2048, 300), (1, 1, 2048, 1152, 1100)])
def test_split_bf16_gemm_is_exact_in_its_operands_and_fp32_accurate(mm, al, bl, M, N, K):
    """gemm_x3.hip takes the large products (nn.Linear / nn.LSTM gate matmuls and their gradients, decoder.py:64-70,106)
    on the bf16 matrix cores with every fp32 operand split exactly into three bf16 pieces.  (a) exactness of the split:
    A times an identity-like B returns A bit for bit (full 24-bit significands, magnitudes from 1e-20 to 1e20);
    (b) accuracy: error against float64 of the size fp32 accumulation gives (<= 4e-6 of max|C| at these K; the fp32
    MFMA kernel measures 0.3-1.2e-6 on the same data)."""
    _lib, ops = mm
    g = torch.Generator().manual_seed(M + N + K + al + 2 * bl)

    def store(logical, k_major):                      # logical [rows, K] -> storage the layout asks for
        return logical.contiguous() if k_major else logical.t().contiguous()

    # (a) B = [I; 0]: C[:, j] = A[:, j] for j < min(N, K), exactly
    scale = 10.0 ** torch.randint(-20, 21, (M, 1), generator=g).float()
    A = torch.randn(M, K, generator=g) * scale
    Bt = torch.zeros(N, K)
    d = min(N, K)
    Bt[torch.arange(d), torch.arange(d)] = 1.0
    As, Bs = dev(store(A, al == 0)), dev(store(Bt, bl == 0))
    C = torch.full((M, N), 7.0, device="cuda")
    ops.gemm(al, bl, M, N, K, As, As.stride(0), Bs, Bs.stride(0), C, N)
    assert torch.equal(C.cpu()[:, :d], A[:, :d]), "the three bf16 pieces do not add up to the fp32 operand"
    assert float(C[:, d:].abs().sum()) == 0.0
    # (b) random operands, bias and accumulation into C
    A = torch.randn(M, K, generator=g)
    Bt = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    want = A.double() @ Bt.double().t() + bias.double() + C0.double()
    As, Bs, C = dev(store(A, al == 0)), dev(store(Bt, bl == 0)), dev(C0)
    ops.gemm(al, bl, M, N, K, As, As.stride(0), Bs, Bs.stride(0), C, N, beta=1, bias=dev(bias))
    close(C, want.float(), tol=4e-6, what=f"split gemm layouts ({al},{bl})")


def test_gemm_rejects_bad_arguments(mm):
    _lib, ops = mm
    x = torch.zeros(4, 4, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm(0, 0, 4, 4, 4, x, 2, x, 4, x, 4)            # lda < K
    with pytest.raises(RuntimeError):
        ops.gemm(0, 5, 4, 4, 4, x, 4, x, 4, x, 4)            # bad layout
    with pytest.raises(RuntimeError):
        ops.linear_fwd(torch.zeros(2, 2), torch.zeros(2, 2), None)   # CPU tensors: no fallback


# ------------------------------------------------------------------------------- embedding
@pytest.mark.parametrize("V,E,n", [(50, 12, 7), (10000, 300, 2048), (97, 301, 33)])
def test_embedding_gather_and_dense_gradient(mm, V, E, n):
    _lib, ops = mm
    g = torch.Generator().manual_seed(V + E)
    table = torch.randn(V, E, generator=g)
    ids = torch.randint(0, V, (n,), generator=g)
    ids[: n // 3] = ids[0]                                    # collisions: the scatter-add must accumulate
    out = ops.embedding_fwd(dev(table), dev(ids))
    assert torch.equal(out.cpu(), table[ids])                # a copy: bit-exact
    dout = torch.randn(n, E, generator=g)
    dtab = torch.zeros(V, E, device="cuda")
    ops.embedding_bwd(dev(dout), dev(ids), dtab)
    want = torch.zeros(V, E, dtype=torch.float64).index_add_(0, ids, dout.double())
    close(dtab, want.float(), what="embedding_bwd")


def test_embedding_out_of_range_ids_give_zero_rows(mm):
    _lib, ops = mm
    table = torch.randn(10, 8)
    ids = torch.tensor([3, -1, 10, 9])
    out = ops.embedding_fwd(dev(table), dev(ids)).cpu()
    assert torch.equal(out[0], table[3]) and torch.equal(out[3], table[9])
    assert out[1].abs().sum() == 0 and out[2].abs().sum() == 0


# ------------------------------------------------------------------------------- attention
def _attn_case(g, B, Lt, Lav, H, Da, Dv):
    S = Lt + 2 * Lav
    scores = torch.randn(B, S, generator=g) * 2
    text = torch.randn(B, Lt, H, generator=g)
    audio = torch.randn(B, Lav, Da, generator=g)
    video = torch.randn(B, Lav, Dv, generator=g)
    text_len = torch.randint(1, Lt + 1, (B,), generator=g).to(torch.int32)
    av_len = torch.randint(1, Lav + 1, (B,), generator=g).to(torch.int32)
    return scores, text, audio, video, text_len, av_len


def _attn_oracle(scores, text, audio, video, text_len, av_len, mask_mode):
    from oracle import mmqg_oracle as O
    Lt, Lav = text.shape[1], audio.shape[1]
    a_t = O._attn_softmax(scores[:, :Lt], text_len, mask_mode)
    a_a = O._attn_softmax(scores[:, Lt:Lt + Lav], av_len, mask_mode)
    a_v = O._attn_softmax(scores[:, Lt + Lav:], av_len, mask_mode)
    ctx = torch.cat((torch.bmm(a_t.unsqueeze(1), text).squeeze(1), torch.bmm(a_a.unsqueeze(1), audio).squeeze(1),
                     torch.bmm(a_v.unsqueeze(1), video).squeeze(1)), dim=1)
    return torch.cat((a_t, a_a, a_v), dim=1), ctx


@pytest.mark.parametrize("shape", [(3, 9, 5, 16, 6, 16), (4, 283, 101, 512, 128, 512), (2, 33, 7, 130, 10, 36),
                                   (64, 32, 8, 512, 128, 512), (1, 283, 101, 512, 128, 512)])
@pytest.mark.parametrize("mask_mode", [0, 1])
def test_attention_forward_backward(mm, shape, mask_mode):
    _lib, ops = mm
    g = torch.Generator().manual_seed(sum(shape) + mask_mode)
    scores, text, audio, video, text_len, av_len = _attn_case(g, *shape)
    leaves = [t.double().requires_grad_(True) for t in (scores, text, audio, video)]
    attn_w, ctx_w = _attn_oracle(*leaves, text_len, av_len, mask_mode)
    dctx = torch.randn(ctx_w.shape, generator=g)
    dattn = torch.randn(attn_w.shape, generator=g) * 0.1
    (attn_w * dattn.double()).sum().add((ctx_w * dctx.double()).sum()).backward()

    ins = [dev(t).requires_grad_(True) for t in (scores, text, audio, video)]
    attn, ctx = ops.AttentionFn.apply(ins[0], ins[1], ins[2], ins[3], dev(text_len), dev(av_len), mask_mode)
    close(attn, attn_w.float(), what="attn weights")
    close(ctx, ctx_w.float(), what="context")
    ((attn * dev(dattn)).sum() + (ctx * dev(dctx)).sum()).backward()
    for name, a, b in zip(("dscores", "dtext", "daudio", "dvideo"), ins, leaves):
        close(a.grad, b.grad.float(), what=name)


@pytest.mark.parametrize("shape", [(3, 9, 5, 16, 8, 16, 32), (64, 283, 101, 512, 128, 512, 512), (5, 33, 7, 132, 12, 36, 100),
                                   (32, 283, 101, 512, 128, 512, 512), (6, 283, 101, 1024, 128, 1024, 1024),
                                   (2, 700, 130, 64, 32, 128, 64)])
@pytest.mark.parametrize("mask_mode,zero_past_len", [(0, 0), (1, 0), (0, 1)])
def test_fused_score_product_softmax_context_in_one_launch(mm, shape, mask_mode, zero_past_len):
    """mmqg_attn_scores_softmax_context_fwd (csrc/attention_fused.hip): scores = pre + h W^T, three softmaxes, three
    contexts in ONE launch — row-range parts with local softmaxes merged by the last arriver of each (question,
    modality) — against the float64 oracle; launched twice on the same workspace (the tickets must be left zero), with
    the no-op and the intended masks and with the zero-padding skip."""
    _lib, ops = mm
    lib = _lib.load()
    B, Lt, Lav, H, Da, Dv, Hq = shape
    g = torch.Generator().manual_seed(sum(shape) + 7 * mask_mode + zero_past_len)
    pre, text, audio, video, text_len, av_len = _attn_case(g, B, Lt, Lav, H, Da, Dv)
    if zero_past_len:
        text = text * (torch.arange(Lt).view(1, -1, 1) < text_len.view(-1, 1, 1))
        audio = audio * (torch.arange(Lav).view(1, -1, 1) < av_len.view(-1, 1, 1))
        video = video * (torch.arange(Lav).view(1, -1, 1) < av_len.view(-1, 1, 1))
    S, Cw = Lt + 2 * Lav, H + Da + Dv
    E = 20                                               # the score matrix is [S][E + Hq]; the kernel gets its recurrent half
    Wfull = torch.randn(S, E + Hq, generator=g) * Hq ** -0.5
    h = torch.randn(B, Hq, generator=g)
    scores = pre.double() + h.double() @ Wfull[:, E:].double().t()
    attn_w, ctx_w = _attn_oracle(scores, text.double(), audio.double(), video.double(), text_len, av_len, mask_mode)
    fused = torch.cat((text.reshape(B, -1), audio.reshape(B, -1), video.reshape(B, -1)), dim=1).cuda()
    v = _lib.AttnValues()
    v.B, v.Lt, v.Lav, v.H, v.Da, v.Dv = B, Lt, Lav, H, Da, Dv
    v.text, v.audio, v.video = fused.data_ptr(), fused.data_ptr() + 4 * Lt * H, fused.data_ptr() + 4 * (Lt * H + Lav * Da)
    v.text_stride_b = v.audio_stride_b = v.video_stride_b = fused.shape[1]
    tl, al = dev(text_len), dev(av_len)
    v.text_len, v.av_len, v.mask_mode, v.zero_past_len = tl.data_ptr(), al.data_ptr(), mask_mode, zero_past_len
    n = int(lib.mmqg_attn_fused_ws_bytes(C.byref(v), Hq))
    assert n > 0, "the fused kernel must take these (16-byte aligned) extents"
    ws = torch.zeros((n + 3) // 4, device="cuda")
    ldS = (S + 3) // 4 * 4
    pre_d = torch.zeros(B, ldS, device="cuda")
    pre_d[:, :S] = pre.cuda()
    Wd, hd = Wfull.cuda(), h.cuda()
    for rep in range(2):
        attn = torch.full((B, ldS), 7.0, device="cuda")
        ctx = torch.full((B, Cw), 7.0, device="cuda")
        rc = lib.mmqg_attn_scores_softmax_context_fwd(C.byref(v), pre_d.data_ptr(), ldS, hd.data_ptr(), Hq,
                                                      Wd.data_ptr() + 4 * E, E + Hq, Hq, attn.data_ptr(), ldS, ctx.data_ptr(), Cw,
                                                      ws.data_ptr(), n, ops._stream())
        assert rc == 0, _lib.load().mmqg_last_error()
        close(attn[:, :S], attn_w.float(), what=f"fused attention weights (launch {rep})")
        close(ctx, ctx_w.float(), what=f"fused contexts (launch {rep})")
        assert bool((attn[:, S:] == 7.0).all()), "wrote past the score row"
    assert bool((ws.view(torch.int32)[-(B * 3):] == 0).all()), "tickets must be zero after a launch"
    # operands the kernel does not take (query rows not 16-byte aligned) are declined, not mangled
    rc = lib.mmqg_attn_scores_softmax_context_fwd(C.byref(v), pre_d.data_ptr(), ldS, hd.data_ptr(), Hq, Wd.data_ptr() + 4 * (E + 1),
                                                  E + Hq, Hq, attn.data_ptr(), ldS, ctx.data_ptr(), Cw, ws.data_ptr(), n, ops._stream())
    assert rc == 1


def test_attention_accepts_the_fused_value_layout(mm):
    """One allocation per question holding text | audio | video rows (the layout the batched
    trainer uses) must give the same result as three separate tensors."""
    _lib, ops = mm
    B, Lt, Lav, H, Da, Dv = 5, 33, 9, 64, 16, 64
    g = torch.Generator().manual_seed(5)
    scores, text, audio, video, text_len, av_len = _attn_case(g, B, Lt, Lav, H, Da, Dv)
    fused = torch.cat((text.reshape(B, -1), audio.reshape(B, -1), video.reshape(B, -1)), dim=1).cuda()
    stride = fused.shape[1]
    v = _lib.AttnValues()
    v.B, v.Lt, v.Lav, v.H, v.Da, v.Dv = B, Lt, Lav, H, Da, Dv
    v.text, v.audio, v.video = fused.data_ptr(), fused.data_ptr() + 4 * Lt * H, fused.data_ptr() + 4 * (Lt * H + Lav * Da)
    v.text_stride_b = v.audio_stride_b = v.video_stride_b = stride
    S, Cw = Lt + 2 * Lav, H + Da + Dv
    sc = scores.cuda()
    attn = torch.empty_like(sc)
    ctx = torch.empty(B, Cw, device="cuda")
    _lib.check(_lib.load().mmqg_attn_softmax_context_fwd(C.byref(v), sc.data_ptr(), S, attn.data_ptr(), S,
                                                         ctx.data_ptr(), Cw, ops._stream()))
    attn_w, ctx_w = _attn_oracle(scores, text, audio, video, text_len, av_len, 0)
    close(attn, attn_w, what="fused attn")
    close(ctx, ctx_w, what="fused ctx")


@pytest.mark.parametrize("mask_mode,zero_past_len", [(0, 0), (1, 0), (0, 1)])
def test_attention_backward_in_one_kernel_equals_the_oracle(mm, mask_mode, zero_past_len):
    """mmqg_attn_context_bwd_fused: the softmax Jacobian's row dot sum_j a_j d(a)_j is taken as ctx . dctx (the
    forward's saved context), so dscores comes out of ONE kernel; against the float64 oracle, also under the
    intended masks and with the zero-padded rows skipped."""
    _lib, ops = mm
    B, Lt, Lav, H, Da, Dv = 6, 41, 13, 96, 24, 64
    g = torch.Generator().manual_seed(17 + mask_mode)
    scores, text, audio, video, text_len, av_len = _attn_case(g, B, Lt, Lav, H, Da, Dv)
    if zero_past_len:                  # the caller's promise: rows past the valid lengths are zero
        for b in range(B):
            text[b, int(text_len[b]):] = 0
            audio[b, int(av_len[b]):] = 0
            video[b, int(av_len[b]):] = 0
    leaves = [t.double().requires_grad_(True) for t in (scores, text, audio, video)]
    attn_w, ctx_w = _attn_oracle(*leaves, text_len, av_len, mask_mode)
    dctx = torch.randn(ctx_w.shape, generator=g)
    (ctx_w * dctx.double()).sum().backward()
    t_d, a_d, v_d, tl_d, al_d = dev(text), dev(audio), dev(video), dev(text_len), dev(av_len)
    v = ops.make_attn_values(t_d, a_d, v_d, tl_d, al_d, mask_mode)
    v.zero_past_len = zero_past_len
    S, Cw = Lt + 2 * Lav, H + Da + Dv
    sc, dc = dev(scores), dev(dctx)
    attn, ctx = torch.empty(B, S, device="cuda"), torch.empty(B, Cw, device="cuda")
    lib = _lib.load()
    _lib.check(lib.mmqg_attn_softmax_context_fwd(C.byref(v), sc.data_ptr(), S, attn.data_ptr(), S, ctx.data_ptr(), Cw, ops._stream()))
    ds = torch.full((B, S), 7.0, device="cuda")
    _lib.check(lib.mmqg_attn_context_bwd_fused(C.byref(v), attn.data_ptr(), S, ctx.data_ptr(), Cw, dc.data_ptr(), Cw,
                                               ds.data_ptr(), S, ops._stream()))
    close(ds, leaves[0].grad.float(), what="dscores from the one-kernel backward")
    ds2 = torch.empty(B, S, device="cuda")                       # and the two-kernel path on the same inputs
    _lib.check(lib.mmqg_attn_context_bwd(C.byref(v), attn.data_ptr(), S, dc.data_ptr(), Cw, None, 0, ds2.data_ptr(), S, ops._stream()))
    close(ds, ds2, tol=1e-5, what="one-kernel vs two-kernel dscores")


def test_attention_dvalues_over_steps(mm):
    _lib, ops = mm
    T, B, Lt, Lav, H, Da, Dv = 6, 4, 11, 5, 32, 8, 16
    S, Cw = Lt + 2 * Lav, H + Da + Dv
    g = torch.Generator().manual_seed(3)
    attn = torch.rand(T, B, S, generator=g)
    dctx = torch.randn(T, B, Cw, generator=g)
    n_rows = 7
    out = torch.full((n_rows, B, H), 7.0, device="cuda")        # time-major [row][b][H]
    a_d, d_d = dev(attn), dev(dctx)                               # keep alive: the call only sees raw pointers
    _lib.check(_lib.load().mmqg_attn_dvalues(T, B, n_rows, H, a_d.data_ptr(), B * S, S, 0, d_d.data_ptr(),
                                             B * Cw, Cw, 0, out.data_ptr(), B * H, H, 0, ops._stream()))
    want = torch.einsum("tbr,tbh->rbh", attn[:, :, :n_rows].double(), dctx[:, :, :H].double())
    close(out, want.float(), what="dvalues text")
    outv = torch.zeros(B, Lav, Dv, device="cuda")
    _lib.check(_lib.load().mmqg_attn_dvalues(T, B, Lav, Dv, a_d.data_ptr(), B * S, S, Lt + Lav, d_d.data_ptr(), B * Cw, Cw,
                                             H + Da, outv.data_ptr(), Dv, Lav * Dv, 0, ops._stream()))
    wantv = torch.einsum("tbr,tbh->brh", attn[:, :, Lt + Lav:].double(), dctx[:, :, H + Da:].double())
    close(outv, wantv.float(), what="dvalues video")


# ------------------------------------------------------------------------------- LSTM cell
@pytest.mark.parametrize("B,H", [(3, 16), (64, 512), (5, 33)])
def test_lstm_cell_forward_backward_with_ragged_rows_and_dropout(mm, B, H):
    _lib, ops = mm
    from oracle import mmqg_oracle as O
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * H)
    pre = torch.randn(B, 4 * H, generator=g)
    h_prev, c_prev = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    lens = torch.randint(0, 4, (B,), generator=g).to(torch.int32)
    t, p, seed, stream = 1, 0.25, 1234, 77
    gates = dev(pre).clone()
    h_out, c_out, h_drop = (torch.empty(B, H, device="cuda") for _ in range(3))
    y = torch.full((B, H + 3), -5.0, device="cuda")
    hp_d, cp_d, lens_d = dev(h_prev), dev(c_prev), dev(lens)     # keep alive: the calls only see raw pointers
    _lib.check(lib.mmqg_lstm_cell_fwd(B, H, gates.data_ptr(), 4 * H, hp_d.data_ptr(), cp_d.data_ptr(),
                                      h_out.data_ptr(), c_out.data_ptr(), h_drop.data_ptr(), y.data_ptr(), H + 3,
                                      lens_d.data_ptr(), t, p, seed, stream, ops._stream()))
    mask = ops.dropout_mask(B * H, p, seed, stream, "cuda").view(B, H).cpu()
    active = (lens > t).view(-1, 1)
    # oracle: the cell with identity "weights" (pre-activations given directly)
    hp, cp, pr = (x.double().requires_grad_(True) for x in (h_prev, c_prev, pre))
    i, f, gg, o = (pr[:, k * H:(k + 1) * H] for k in range(4))
    c_new = torch.sigmoid(f) * cp + torch.sigmoid(i) * torch.tanh(gg)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    h_w = torch.where(active, h_new, hp)
    c_w = torch.where(active, c_new, cp)
    close(h_out, h_w.float(), what="h")
    close(c_out, c_w.float(), what="c")
    close(h_drop, (h_w * mask.double() * active).float(), what="h_drop")
    close(y[:, :H], (h_w * active).float(), what="y")
    assert torch.all(y[:, H:] == -5.0)
    act = torch.cat((torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)), dim=1) * active
    close(gates, act.float(), what="activated gates")
    # backward: loss = sum(h_drop*gd) + sum(y*gy) + sum(h*gh) + sum(c*gc)
    gd, gy, gh, gc = (torch.randn(B, H, generator=g) for _ in range(4))
    loss = (h_w * mask.double() * active * gd.double()).sum() + (h_w * active * gy.double()).sum() \
        + (h_w * gh.double()).sum() + (c_w * gc.double()).sum()
    loss.backward()
    dh_rec = dev(gh).clone()
    dc = dev(gc).clone()
    dgates = torch.empty(B, 4 * H, device="cuda")
    gd_d, gy_d = dev(gd), dev(gy)
    _lib.check(lib.mmqg_lstm_cell_bwd(B, H, gates.data_ptr(), cp_d.data_ptr(), c_out.data_ptr(), dh_rec.data_ptr(),
                                      gd_d.data_ptr(), H, p, seed, stream, gy_d.data_ptr(), H, dc.data_ptr(),
                                      dgates.data_ptr(), 4 * H, lens_d.data_ptr(), t, ops._stream()))
    close(dgates, pr.grad.float(), what="dgates")
    close(dc, cp.grad.float(), what="dc_prev")
    # dh_rec holds only the pass-through part (finished rows); active rows get theirs from dgates*W_hh
    close(dh_rec, (hp.grad * (~active)).float(), what="dh pass-through")


def test_dropout_mask_statistics_and_determinism(mm):
    _lib, ops = mm
    n, p = 1 << 20, 0.2
    a = ops.dropout_mask(n, p, 9, 3, "cuda")
    b = ops.dropout_mask(n, p, 9, 3, "cuda")
    c = ops.dropout_mask(n, p, 9, 4, "cuda")
    off = torch.tensor([5], dtype=torch.int32, device="cuda")
    d = ops.dropout_mask(n, p, 9, 3, "cuda", seed_offset=off)
    assert torch.equal(a, b)
    assert not torch.equal(a, c) and not torch.equal(a, d)
    vals = sorted(a.unique().cpu().tolist())
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.25) < 1e-6
    keep = float((a > 0).float().mean())
    assert abs(keep - 0.8) < 4 * (0.8 * 0.2 / n) ** 0.5 + 1e-3
    assert abs(float(a.mean()) - 1.0) < 5e-3
    assert abs(float(((a > 0) & (c > 0)).float().mean()) - 0.64) < 5e-3      # streams independent


# ------------------------------------------------------------------- cross entropy / sums / Adam
@pytest.mark.parametrize("rows,V", [(7, 50), (1280, 10000), (3, 50001)])
def test_cross_entropy_argmax_and_gradient(mm, rows, V):
    _lib, ops = mm
    g = torch.Generator().manual_seed(rows + V)
    logits = torch.randn(rows, V, generator=g) * 3
    target = torch.randint(0, V, (rows,), generator=g)
    wgt = torch.rand(rows, generator=g)
    wgt[0] = 0.0
    logits[1, 5] = logits[1, 9] = logits[1].max() + 1       # tie: first index must win
    loss_rows, argmax, dlog = ops.ce_fwd_bwd(dev(logits), dev(target), dev(wgt), want_grad=True)
    ld = logits.double().requires_grad_(True)
    ce = torch.nn.functional.cross_entropy(ld, target, reduction="none") * wgt.double()
    ce.sum().backward()
    close(loss_rows, ce.float(), what="loss rows")
    close(dlog, ld.grad.float(), what="dlogits")
    assert torch.equal(argmax.cpu(), torch.argmax(logits, dim=1))
    assert int(argmax[1]) == 5
    # in place
    buf = dev(logits).clone()
    _, _, d2 = ops.ce_fwd_bwd(buf, dev(target), dev(wgt), want_grad=True, in_place=True)
    assert d2.data_ptr() == buf.data_ptr()
    close(buf, ld.grad.float(), what="dlogits in place")


@pytest.mark.parametrize("rows,V,H,ld,tiled", [(1280, 10000, 512, None, True), (1270, 7777, 96, 7780, True),
                                                 (2560, 12000, 128, 12032, True), (1270, 7777, 96, None, False),
                                                 (1270, 7777, 512, None, True), (300, 5001, 640, 5004, True),
                                                 (80, 10000, 512, None, False)])
def test_projection_with_loss_statistics_equals_the_three_sweep_loss(mm, rows, V, H, ld, tiled):
    """decoder.py:106 + train.py:174: the projection's epilogue hands max / sum-exp / first argmax per column tile to
    the loss kernel; loss, argmax (ties: first index, within a lane, across lanes and across tiles) and gradient must
    equal the kernel that sweeps the logits itself, and the float64 loss of the same logits."""
    _lib, ops = mm
    g = torch.Generator().manual_seed(rows * 7 + V)
    h = torch.randn(rows, H, generator=g)
    W = torch.randn(V, H, generator=g) / H ** 0.5
    b = torch.randn(V, generator=g) * 0.1
    far = min(V - 1, 5 + 16 * 13 * 3)
    W[5] = W[9] = W[6] = W[far] = 6 * h[1] / h[1].norm()            # one maximum, four times, in row 1
    b[5] = b[9] = b[6] = b[far] = 0.25
    target = torch.randint(0, V, (rows,), generator=g)
    wgt = torch.rand(rows, generator=g)
    wgt[0] = 0.0
    logits, stats, tiles = ops.projection_fwd(dev(h), dev(W), dev(b), ld=ld)
    assert (tiles > 0) == tiled          # (few rows, or an odd row pitch on the fp32 kernels: generic kernel, no statistics)
    want_logits = h.double() @ W.double().T + b.double()
    close(logits, want_logits.float(), what="projection logits")
    keep = logits.clone()
    loss3, arg3, d3 = ops.ce_fwd_bwd(keep.clone(), dev(target), dev(wgt), want_grad=True)
    loss1, arg1, d1 = ops.ce_fwd_bwd(logits, dev(target), dev(wgt), want_grad=True, in_place=True, stats=stats,
                                     stats_tiles=tiles)
    assert d1.data_ptr() == logits.data_ptr()
    assert torch.equal(arg1, arg3) and int(arg1[1]) == 5
    assert torch.equal(arg1.cpu(), torch.argmax(keep.cpu(), dim=1))
    ld = keep.cpu().double().requires_grad_(True)
    ce = torch.nn.functional.cross_entropy(ld, target, reduction="none") * wgt.double()
    ce.sum().backward()
    close(loss1, ce.float(), tol=2e-6, what="loss rows from statistics")
    close(d1, ld.grad.float(), tol=2e-6, what="dlogits from statistics")
    close(loss1, loss3, tol=2e-6, what="loss rows, one sweep against three")
    close(d1, d3, tol=2e-6, what="dlogits, one sweep against three")


@pytest.mark.parametrize("rows,out_f,in_f", [(1280, 10000, 512), (1000, 2049, 520), (7, 50, 12), (600, 300, 96)])
def test_linear_weight_and_bias_gradient_in_one_product(mm, rows, out_f, in_f):
    """Autograd of decoder.py:106 (nn.Linear): dW += dY^T X and db += column sums of dY.  On large shapes the column
    sums are taken from the registers of the product's staging pass (csrc/gemm_x3.hip), on small ones by a sweep."""
    _lib, ops = mm
    g = torch.Generator().manual_seed(rows + out_f)
    dY = torch.randn(rows, out_f, generator=g) * 0.1
    X = torch.randn(rows, in_f, generator=g)
    dW0 = torch.randn(out_f, in_f, generator=g)
    db0 = torch.randn(out_f, generator=g)
    dW, db = dev(dW0), dev(db0)
    ops.linear_wgrad(dev(dY), dev(X), dW, db)
    close(dW, (dW0.double() + dY.double().t() @ X.double()).float(), tol=4e-6, what="dW")
    close(db, (db0.double() + dY.double().sum(0)).float(), tol=4e-6, what="db")
    dW2 = dev(dW0)
    ops.linear_wgrad(dev(dY), dev(X), dW2, None)                     # no bias gradient asked for
    close(dW2, dW, tol=1e-6, what="dW without db")


def test_colsum_and_reduce_sum(mm):
    _lib, ops = mm
    g = torch.Generator().manual_seed(1)
    X = torch.randn(1280, 2048, generator=g)
    out = torch.ones(2048, device="cuda")
    ops.colsum_add(dev(X), out)
    close(out, (X.double().sum(0) + 1).float(), what="colsum")
    # unaligned width / padded rows (scalar kernel), a sub-block of a wider matrix (vector kernel, ld > N), few rows
    for M, N, ld in ((37, 485, 488), (5, 64, 64), (130, 300, 812), (1, 4, 4)):
        Xp = torch.randn(M, ld, generator=g)
        o = torch.zeros(N, device="cuda")
        Xd = dev(Xp)
        _lib.check(_lib.load().mmqg_colsum_add(Xd.data_ptr(), ld, M, N, o.data_ptr(), ops._stream()))
        close(o, Xp[:, :N].double().sum(0).float(), what=f"colsum {M}x{N} ld {ld}")
    x = torch.randn(1000, generator=g)
    r = torch.zeros(1, device="cuda")
    x_d = dev(x)
    _lib.check(_lib.load().mmqg_reduce_sum(x_d.data_ptr(), 1000, r.data_ptr(), ops._stream()))
    close(r, x.double().sum().float().view(1), what="reduce_sum")


def test_adam_matches_oracle_over_steps(mm):
    _lib, ops = mm
    from oracle import mmqg_oracle as O
    lib = _lib.load()
    n = 10007
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(n, generator=g)
    p_ref, m_ref, v_ref = p0.clone(), torch.zeros(n), torch.zeros(n)
    pad = (n + 3) // 4 * 4
    p, m, v = torch.zeros(pad, device="cuda"), torch.zeros(pad, device="cuda"), torch.zeros(pad, device="cuda")
    p[:n] = dev(p0)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for it in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** (it - 2))
        O.adam_update(p_ref, grad, m_ref, v_ref, it, lr=1e-4)
        gd = torch.zeros(pad, device="cuda")
        gd[:n] = dev(grad) * 4.0                              # grad_scale 0.25 undoes this
        _lib.check(lib.mmqg_counter_add(step.data_ptr(), 1, ops._stream()))
        _lib.check(lib.mmqg_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8,
                                      step.data_ptr(), 0.25, ops._stream()))
        close(p[:n], p_ref, tol=2e-7, what=f"adam p step {it}")
        close(m[:n], m_ref, tol=1e-6, what="adam m")
        close(v[:n], v_ref, tol=1e-6, what="adam v")
    assert int(step) == 3


# ------------------------------------------------------------------------ sequence executors
@pytest.mark.parametrize("T,B,L,H,In,p", [(5, 3, 3, 16, 12, 0.0), (4, 6, 2, 32, 20, 0.3), (7, 4, 1, 24, 40, 0.0),
                                          # shapes the persistent forward time loop takes (csrc/persist.hip): partial
                                          # row blocks, 1..4 row blocks per unit, 1..3 layers, dropout on and off
                                          (6, 5, 3, 128, 40, 0.25), (5, 20, 2, 128, 64, 0.0), (4, 64, 1, 256, 32, 0.0),
                                          (9, 33, 3, 192, 24, 0.3), (3, 48, 2, 512, 16, 0.2),
                                          # the bench's own text-encoder shape and a ragged three-layer one for the
                                          # persistent BACKWARD loop (K-sliced weights, two barrier phases per diagonal)
                                          (32, 64, 3, 512, 300, 0.2), (11, 37, 3, 256, 40, 0.25), (2, 1, 1, 128, 8, 0.0),
                                          # T*B = 1536 rows: the weight gradients go through the split-bf16 GEMM group, the
                                          # bias gradients ride in its staging pass (csrc/gemm_x3.hip); T*B not a multiple of 32
                                          (24, 64, 2, 256, 64, 0.2), (17, 45, 2, 256, 48, 0.0),
                                          # more than 64 rows: the wide forward layer-step kernel (64 rows x 8 units per
                                          # workgroup, csrc/skinny.hip), with partial row and unit tiles
                                          (3, 128, 2, 1024, 64, 0.2), (2, 100, 1, 1036, 40, 0.0), (4, 70, 3, 264, 24, 0.3)])
def test_lstm_sequence_executor_forward_backward(mm, T, B, L, H, In, p):
    """mmqg_lstm_seq_fwd/bwd (autograd wrapper LSTMSeqFn) vs the oracle's stacked cell loop, with
    a given initial state and the executor's own dropout masks replayed into the oracle."""
    _lib, ops = mm
    from oracle import mmqg_oracle as O
    persistent_before = _lib.load().mmqg_persist_launch_count()
    persistent_bwd_before = _lib.load().mmqg_persist_bwd_launch_count()
    g = torch.Generator().manual_seed(T * B + H)
    params = {}
    for l in range(L):
        d = In if l == 0 else H
        params[f"lstm.weight_ih_l{l}"] = torch.randn(4 * H, d, generator=g) * d ** -0.5
        params[f"lstm.weight_hh_l{l}"] = torch.randn(4 * H, H, generator=g) * H ** -0.5
        params[f"lstm.bias_ih_l{l}"] = torch.randn(4 * H, generator=g) * 0.3
        params[f"lstm.bias_hh_l{l}"] = torch.randn(4 * H, generator=g) * 0.3
    x = torch.randn(T, B, In, generator=g)
    h0, c0 = torch.randn(L, B, H, generator=g) * 0.5, torch.randn(L, B, H, generator=g) * 0.5
    gy, ghT, gcT = torch.randn(T, B, H, generator=g), torch.randn(L, B, H, generator=g), torch.randn(L, B, H, generator=g)
    seed = 4242

    flat = [dev(params[f"lstm.{n}_l{l}"]).requires_grad_(True) for l in range(L)
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    xd, h0d, c0d = (dev(t).requires_grad_(True) for t in (x, h0, c0))
    y, hT, cT = ops.lstm_seq(xd, h0d, c0d, flat, p, True, seed)
    took_persistent = _lib.load().mmqg_persist_launch_count() - persistent_before
    assert took_persistent == (1 if H >= 128 and B <= 64 else 0), "the persistent time loop must take exactly the wide shapes"
    ((y * dev(gy)).sum() + (hT * dev(ghT)).sum() + (cT * dev(gcT)).sum()).backward()
    took_persistent_bwd = _lib.load().mmqg_persist_bwd_launch_count() - persistent_bwd_before
    assert took_persistent_bwd == (1 if H >= 128 and H % 64 == 0 and B <= 64 else 0), \
        "the persistent backward time loop must take exactly the wide shapes (csrc/persist_bwd.hip)"
    # replay the executor's masks: stream id = stream_base + l*T + t, element = b*H + j
    base = (next(ops._stream_counter) - 1) << 32
    masks = None
    if p > 0 and L > 1:
        masks = [[ops.dropout_mask(B * H, p, seed, base + l * T + t, "cuda").view(B, H).cpu().double()
                  for l in range(L - 1)] for t in range(T)]

    pd = {k: v.double().requires_grad_(True) for k, v in params.items()}
    xo, h0o, c0o = (t.double().requires_grad_(True) for t in (x, h0, c0))
    hid = (h0o, c0o)
    outs = []
    for t in range(T):
        out, hid = O.lstm_stack_step(pd, "lstm.", L, xo[t], hid, None if masks is None else masks[t])
        outs.append(out)
    yo = torch.stack(outs)
    ((yo * gy.double()).sum() + (hid[0] * ghT.double()).sum() + (hid[1] * gcT.double()).sum()).backward()
    close(y, yo.float(), what="y")
    close(hT, hid[0].float(), what="hT")
    close(cT, hid[1].float(), what="cT")
    close(xd.grad, xo.grad.float(), what="dx")
    close(h0d.grad, h0o.grad.float(), what="dh0")
    close(c0d.grad, c0o.grad.float(), what="dc0")
    i = 0
    for l in range(L):
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            close(flat[i].grad, pd[f"lstm.{n}_l{l}"].grad.float(), what=f"d{n}_l{l}")
            i += 1


def test_persistent_time_loop_is_bitwise_repeatable_under_load(mm):
    """The persistent forward hands h(t) between workgroups through write-through stores, sc1 loads and a fence-free
    device-wide barrier (csrc/grid_barrier.h).  A stale read would show up as a run that differs from the others: the
    same sequence is run 150 times — half of them beside a bandwidth-heavy GEMM stream on another queue (uneven
    load, the regime in which broken hand-offs surface) — and every output must be bit-identical to the first run's,
    which in turn matches the launch-per-diagonal path."""
    _lib, ops = mm
    lib = _lib.load()
    T, B, L, H, In = 12, 64, 3, 256, 64
    g = torch.Generator().manual_seed(99)
    params = []
    for l in range(L):
        d = In if l == 0 else H
        params += [torch.randn(4 * H, d, generator=g) * d ** -0.5, torch.randn(4 * H, H, generator=g) * H ** -0.5,
                   torch.randn(4 * H, generator=g) * 0.3, torch.randn(4 * H, generator=g) * 0.3]
    params = [dev(p) for p in params]
    x, h0, c0 = dev(torch.randn(T, B, In, generator=g)), dev(torch.randn(L, B, H, generator=g) * 0.5), dev(torch.randn(L, B, H, generator=g) * 0.5)

    def run():
        with torch.no_grad():
            y, hT, cT = ops.lstm_seq(x, h0, c0, params, 0.3, True, 4321)
        return torch.cat((y.reshape(-1), hT.reshape(-1), cT.reshape(-1)))

    before = lib.mmqg_persist_launch_count()
    ops._stream_counter = __import__("itertools").count(7)          # the same dropout streams for every run
    first = run()
    assert lib.mmqg_persist_launch_count() == before + 1
    side = torch.cuda.Stream()
    a_big, b_big = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
    for i in range(150):
        if i % 2:
            with torch.cuda.stream(side):
                for _ in range(3):
                    a_big @ b_big
        ops._stream_counter = __import__("itertools").count(7)
        out = run()
        assert torch.equal(out, first), f"run {i} differs from the first run (max abs diff {float((out - first).abs().max()):.3e})"
    torch.cuda.synchronize()
    # the launch-per-diagonal path on the same inputs (different summation order inside a product: not bitwise)
    ops._stream_counter = __import__("itertools").count(7)
    ws_bytes = lib.mmqg_lstm_seq_persist_ws_bytes
    try:
        lib.mmqg_lstm_seq_persist_ws_bytes = lambda *a: 0              # no workspace -> the executor takes the launches
        ref = run()
    finally:
        lib.mmqg_lstm_seq_persist_ws_bytes = ws_bytes
    close(first, ref, what="persistent vs launch-per-diagonal outputs")


def test_persistent_backward_is_bitwise_repeatable_under_load(mm):
    """The persistent backward hands gate gradients and partial product tiles between workgroups through write-through
    stores, sc1 loads and two fence-free device-wide barriers per anti-diagonal.  A stale read would show up as a run
    that differs: the same backward runs 100 times, half of them beside a GEMM stream on another queue, and every
    gradient must be bit-identical to the first run's, which in turn matches the launch-per-diagonal path."""
    _lib, ops = mm
    lib = _lib.load()
    T, B, L, H, In = 10, 64, 3, 256, 64
    g = torch.Generator().manual_seed(123)
    params = []
    for l in range(L):
        d = In if l == 0 else H
        params += [torch.randn(4 * H, d, generator=g) * d ** -0.5, torch.randn(4 * H, H, generator=g) * H ** -0.5,
                   torch.randn(4 * H, generator=g) * 0.3, torch.randn(4 * H, generator=g) * 0.3]
    params = [dev(p).requires_grad_(True) for p in params]
    x = dev(torch.randn(T, B, In, generator=g)).requires_grad_(True)
    h0 = dev(torch.randn(L, B, H, generator=g) * 0.5).requires_grad_(True)
    c0 = dev(torch.randn(L, B, H, generator=g) * 0.5).requires_grad_(True)
    gy, ghT, gcT = (dev(torch.randn(*shape, generator=g)) for shape in ((T, B, H), (L, B, H), (L, B, H)))

    def run():
        ops._stream_counter = __import__("itertools").count(11)
        for t in [x, h0, c0] + params:
            t.grad = None
        y, hT, cT = ops.lstm_seq(x, h0, c0, params, 0.3, True, 977)
        ((y * gy).sum() + (hT * ghT).sum() + (cT * gcT).sum()).backward()
        return torch.cat([t.grad.reshape(-1) for t in [x, h0, c0] + params])

    before = lib.mmqg_persist_bwd_launch_count()
    first = run()
    assert lib.mmqg_persist_bwd_launch_count() == before + 1
    side = torch.cuda.Stream()
    a_big, b_big = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
    for i in range(100):
        if i % 2:
            with torch.cuda.stream(side):
                for _ in range(3):
                    a_big @ b_big
        out = run()
        # dc0 is the kernel's own output (the cell-state gradient carried through every step, a function of every
        # product of the loop): bit-identical.  dx, dh0 and the weight gradients come out of k-sliced GEMMs whose
        # atomic sums depend on arrival order in the last bits: held to 1e-5 of their magnitude.
        lo = x.numel() + h0.numel()
        n_exact = lo + c0.numel()
        assert torch.equal(out[lo:n_exact], first[lo:n_exact]), \
            f"run {i}: dc0 differs from the first run (max abs diff {float((out[lo:n_exact] - first[lo:n_exact]).abs().max()):.3e})"
        close(out[:lo], first[:lo], tol=1e-5, what=f"dx, dh0 of run {i}")
        close(out[n_exact:], first[n_exact:], tol=1e-5, what=f"weight gradients of run {i}")
    torch.cuda.synchronize()
    ws_bytes = lib.mmqg_lstm_seq_bwd_persist_ws_bytes
    try:
        lib.mmqg_lstm_seq_bwd_persist_ws_bytes = lambda *a: 0            # no workspace -> one launch per anti-diagonal
        ref = run()
    finally:
        lib.mmqg_lstm_seq_bwd_persist_ws_bytes = ws_bytes
    assert lib.mmqg_persist_bwd_launch_count() == before + 101
    close(first, ref, tol=2e-5, what="persistent vs launch-per-diagonal gradients")


def test_a_failed_persistent_launch_is_loud_and_a_concurrent_request_is_declined(mm):
    """(1) Two persistent time loops must not be in flight on one device: a request on another stream while the
    previous launch has not completed is declined and takes the launch-per-diagonal path (same results).
    (2) A launch whose device-wide barrier cannot complete (test hook: it waits for one workgroup more than the grid
    has, with a short spin bound) poisons its output with NaN, sets the health word — the next executor call raises
    — and after mmqg_persist_clear_failures() everything works again."""
    _lib, ops = mm
    lib = _lib.load()
    T, B, L, H, In = 6, 32, 2, 256, 48
    g = torch.Generator().manual_seed(5)
    params = []
    for l in range(L):
        d = In if l == 0 else H
        params += [torch.randn(4 * H, d, generator=g) * d ** -0.5, torch.randn(4 * H, H, generator=g) * H ** -0.5,
                   torch.randn(4 * H, generator=g) * 0.3, torch.randn(4 * H, generator=g) * 0.3]
    params = [dev(p) for p in params]
    x, h0, c0 = dev(torch.randn(T, B, In, generator=g)), dev(torch.zeros(L, B, H)), dev(torch.zeros(L, B, H))

    def run():
        ops._stream_counter = __import__("itertools").count(3)
        with torch.no_grad():
            return ops.lstm_seq(x, h0, c0, params, 0.0, True, 1)[0]

    ref = run()
    torch.cuda.synchronize()
    # (1) keep stream A busy in front of its persistent launch, then ask from stream B at once
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    a_big = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    declined0, launched0 = lib.mmqg_persist_declined_count(), lib.mmqg_persist_launch_count()
    with torch.cuda.stream(sa):
        for _ in range(20):
            a_big @ a_big                        # tens of milliseconds
        ya = run()
    with torch.cuda.stream(sb):
        yb = run()
    torch.cuda.synchronize()
    assert lib.mmqg_persist_launch_count() == launched0 + 1, "the second request must not have been launched persistently"
    assert lib.mmqg_persist_declined_count() == declined0 + 1
    assert torch.equal(ya, ref)
    close(yb, ref, what="declined request on the launch-per-diagonal path")
    # (2) the failure path
    assert lib.mmqg_persist_failures() == 0
    try:
        lib.mmqg_persist_set_test_fault(1, 2048)
        bad = run()
        torch.cuda.synchronize()
        assert lib.mmqg_persist_failures() > 0, "a timed-out barrier must reach the host"
        assert not bool(torch.isfinite(bad).all()), "the failed launch must poison its output"
        with pytest.raises(_lib.BackendError, match="timed out"):
            run()
    finally:
        lib.mmqg_persist_set_test_fault(0, 0)
        lib.mmqg_persist_clear_failures()
    again = run()
    torch.cuda.synchronize()
    assert lib.mmqg_persist_failures() == 0 and torch.equal(again, ref)


def test_transpose(mm):
    _lib, ops = mm
    g = torch.Generator().manual_seed(8)
    src = torch.randn(77, 130 + 6, generator=g)
    s_d = dev(src)
    dst = torch.full((130, 80), -1.0, device="cuda")
    _lib.check(_lib.load().mmqg_transpose_f32(s_d.data_ptr(), 136, 77, 130, dst.data_ptr(), 80, ops._stream()))
    assert torch.equal(dst[:, :77].cpu(), src[:, :130].t())
    assert torch.all(dst[:, 77:] == -1.0)
    # several matrices of different shapes in one launch (17 jobs: more than one batch of 16)
    shapes = [(77, 130, 6, 3), (2048, 512, 0, 0), (5, 3, 1, 2), (33, 64, 0, 31)] * 4 + [(1, 1, 0, 0)]
    jobs = (_lib.TransposeJob * len(shapes))()
    keep = []
    for j, (r, c, ps, pd) in zip(jobs, shapes):
        a = torch.randn(r, c + ps, generator=g)
        a_d, d_d = dev(a), torch.full((c, r + pd), -2.0, device="cuda")
        keep.append((a, a_d, d_d, r, c))
        j.src, j.ld_src, j.rows, j.cols, j.dst, j.ld_dst = a_d.data_ptr(), c + ps, r, c, d_d.data_ptr(), r + pd
    _lib.check(_lib.load().mmqg_transpose_f32_batch(jobs, len(shapes), ops._stream()))
    torch.cuda.synchronize()
    for a, a_d, d_d, r, c in keep:
        assert torch.equal(d_d[:, :r].cpu(), a[:, :c].t()) and torch.all(d_d[:, r:] == -2.0)
    # a batch in which every job is 16-byte aligned with extents that are multiples of 4: the 64 x 64 float4 kernel (the
    # weight matrices of the model: [4H][H], [4H][H + Da + Dv] sub-blocks with a leading dimension of their own, partial
    # tiles at the edges)
    shapes = [(2048, 512, 0, 0), (2048, 1152, 300, 0), (485, 512, 300, 3), (101, 36, 4, 7), (4, 4, 0, 0), (3, 8, 0, 1)]
    jobs = (_lib.TransposeJob * len(shapes))()
    keep = []
    for j, (r, c, ps, pd) in zip(jobs, shapes):
        a = torch.randn(r, c + ps, generator=g)
        a_d, d_d = dev(a), torch.full((c, r + pd), -2.0, device="cuda")
        keep.append((a, a_d, d_d, r, c, ps))
        j.src, j.ld_src, j.rows, j.cols, j.dst, j.ld_dst = a_d.data_ptr() + 4 * ps, c + ps, r, c, d_d.data_ptr(), r + pd
    _lib.check(_lib.load().mmqg_transpose_f32_batch(jobs, len(shapes), ops._stream()))
    torch.cuda.synchronize()
    for a, a_d, d_d, r, c, ps in keep:
        assert torch.equal(d_d[:, :r].cpu(), a[:, ps:ps + c].t()) and torch.all(d_d[:, r:] == -2.0)


# ------------------------------------------------------------------------------ edge cases
def test_empty_and_degenerate_shapes(mm):
    _lib, ops = mm
    lib = _lib.load()
    x = torch.zeros(4, 4, device="cuda")
    ops.gemm(0, 0, 0, 4, 4, x, 4, x, 4, x, 4)                     # M = 0: nothing to do
    ops.gemm(0, 0, 4, 0, 4, x, 4, x, 4, x, 4)                     # N = 0
    out = ops.embedding_fwd(torch.randn(5, 8, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda"))
    assert out.shape == (0, 8)
    _lib.check(lib.mmqg_ce_fwd_bwd(x.data_ptr(), 4, None, None, 0, 4, None, None, None, 0, ops._stream()))
    # single question, single value row per modality
    g = torch.Generator().manual_seed(0)
    sc, text, audio, video, tl, al = _attn_case(g, 1, 1, 1, 8, 4, 8)
    attn, ctx = ops.AttentionFn.apply(sc.cuda(), text.cuda(), audio.cuda(), video.cuda(), None, None, 0)
    assert torch.allclose(attn.cpu(), torch.ones(1, 3))          # softmax over one slot
    close(ctx, torch.cat((text[0, 0], audio[0, 0], video[0, 0]))[None], what="single-row context")


def test_sequence_with_rows_that_never_start(mm):
    """lens[b] == 0: the row keeps its initial state through the whole sequence, its outputs are
    zero rows (what train.py:160's zero-initialised all_enc_outputs holds) and it gets no gradient."""
    _lib, ops = mm
    lib = _lib.load()
    T, B, L, H, In = 4, 3, 2, 16, 8
    g = torch.Generator().manual_seed(1)
    params = []
    for l in range(L):
        d = In if l == 0 else H
        params += [torch.randn(4 * H, d, generator=g) * 0.3, torch.randn(4 * H, H, generator=g) * 0.3,
                   torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1]
    params = [p.cuda() for p in params]
    x = torch.randn(T, B, In, generator=g).cuda()
    h0 = torch.randn(L, B, H, generator=g).cuda()
    c0 = torch.randn(L, B, H, generator=g).cuda()
    lens = torch.tensor([0, 4, 2], dtype=torch.int32, device="cuda")
    d = _lib.LstmSeq()
    d.T, d.B, d.L, d.H, d.In = T, B, L, H, In
    d.x, d.ldx = x.data_ptr(), In
    for l in range(L):
        d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = (p.data_ptr() for p in params[4 * l:4 * l + 4])
    d.h0, d.c0, d.lens = h0.data_ptr(), c0.data_ptr(), lens.data_ptr()
    gates = torch.empty(L, T, B, 4 * H, device="cuda")
    hs, cs = torch.empty(L, T + 1, B, H, device="cuda"), torch.empty(L, T + 1, B, H, device="cuda")
    y = torch.full((T, B, H), 9.0, device="cuda")
    d.gates, d.hs, d.cs = gates.data_ptr(), hs.data_ptr(), cs.data_ptr()
    d.y, d.y_stride_t, d.y_stride_b = y.data_ptr(), B * H, H
    _lib.check(lib.mmqg_lstm_seq_fwd(C.byref(d), ops._stream()))
    assert torch.equal(hs[:, T, 0], h0[:, 0]) and torch.equal(cs[:, T, 0], c0[:, 0])      # never started: state carried
    assert float(y[:, 0].abs().sum()) == 0 and float(y[2:, 2].abs().sum()) == 0           # zero rows past the length
    assert float(y[:2, 2].abs().sum()) > 0 and float(gates[:, :, 0].abs().sum()) == 0
    assert torch.equal(hs[:, T, 2], hs[:, 2, 2])                                           # row 2 frozen after 2 steps


# ------------------------------------------------------------------------------- frame CNN
def _cnn_params(cin, seed, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    vid, chans = {}, [cin, 4, 6, 8, 10]
    for i in range(1, 5):
        ci, co = chans[i - 1], chans[i]
        vid[f"conv{i}.weight"] = (torch.randn(co, ci, 3, 3, generator=g, dtype=dtype) / (3.0 * ci ** 0.5))
        vid[f"conv{i}.bias"] = torch.randn(co, generator=g, dtype=dtype) * 0.1
        vid[f"bn{i}.weight"] = 1.0 + 0.3 * torch.randn(co, generator=g, dtype=dtype)     # some gammas may be < 0
        vid[f"bn{i}.bias"] = 0.2 * torch.randn(co, generator=g, dtype=dtype)
        vid[f"bn{i}.running_mean"] = 0.1 * torch.randn(co, generator=g, dtype=dtype)
        vid[f"bn{i}.running_var"] = 0.5 + torch.rand(co, generator=g, dtype=dtype)
    vid["bn2.weight"][0] = -0.7                                                         # pooling after a negative scale
    return vid


def _cnn_hip(ops, vid, frames, n_frames, training):
    params = []
    for i in range(1, 5):
        params += [vid[f"conv{i}.weight"], vid[f"conv{i}.bias"], vid[f"bn{i}.weight"], vid[f"bn{i}.bias"],
                   vid[f"bn{i}.running_mean"], vid[f"bn{i}.running_var"]]
    return ops.FrameCNNFn.apply(frames, n_frames, training, 1e-5, 0.1, (False, True, False, True), *params)


@pytest.mark.parametrize("B,T,Cin,HW,ragged", [(3, 5, 3, 40, True), (2, 4, 3, 56, False), (1, 3, 2, 41, True),
                                               (4, 8, 3, 112, True)])
def test_frame_cnn_forward_backward_and_running_stats(mm, B, T, Cin, HW, ragged):
    """conv3x3 -> ReLU -> per-question BatchNorm (-> max-pool) x4 (encoder.py:40-50,64-67) against the
    float64 oracle: features, every parameter gradient, the running statistics after the step."""
    _lib, ops = mm
    from oracle import mmqg_oracle as O
    g = torch.Generator().manual_seed(B * 100 + HW)
    frames = torch.randn(B, T, Cin, HW, HW, generator=g, dtype=torch.float64)
    n_frames = torch.tensor([max(1, T - 2 * b) for b in range(B)]) if ragged else torch.full((B,), T)
    valid = (torch.arange(T).view(1, -1) < n_frames.view(-1, 1))
    frames = frames * valid.view(B, T, 1, 1, 1)                                          # padding frames are zeros
    vid64 = _cnn_params(Cin, 7)
    leaf = {k: v.clone().requires_grad_(not k.endswith(("running_mean", "running_var"))) for k, v in vid64.items()}
    want = O.frame_cnn(leaf, frames, n_frames, True)
    dfeat = torch.randn(want.shape, generator=g, dtype=torch.float64) * valid.view(B, T, 1)
    (want * dfeat).sum().backward()

    vid32 = {k: dev(v.float()).requires_grad_(not k.endswith(("running_mean", "running_var"))) for k, v in vid64.items()}
    fr32, nf = dev(frames.float()), dev(n_frames.int())
    got = _cnn_hip(ops, vid32, fr32, nf, True)
    got_flat = got.reshape(B, T, -1)
    (got_flat * dev(dfeat.float())).sum().backward()
    torch.cuda.synchronize()
    close(got_flat * dev(valid.view(B, T, 1).float()), want.detach() * valid.view(B, T, 1), what="features")
    assert float(got_flat.detach()[~dev(valid)].abs().max() if (~valid).any() else 0.0) == 0.0    # padding frames -> zero rows
    for k, v in leaf.items():
        if v.requires_grad:
            close(vid32[k].grad, v.grad, tol=2e-4, what=f"d{k}")
        else:
            close(vid32[k], v, what=k)                                                   # running stats after B questions

    # eval mode: normalise with the (advanced) running statistics, nothing is updated
    before = {k: v.clone() for k, v in vid32.items() if "running" in k}
    with torch.no_grad():
        got_e = _cnn_hip(ops, vid32, fr32, nf, False).reshape(B, T, -1)
    want_e = O.frame_cnn({k: v.detach() for k, v in leaf.items()}, frames, n_frames, False)
    close(got_e * dev(valid.view(B, T, 1).float()), want_e * valid.view(B, T, 1), what="eval features")
    for k, v in before.items():
        assert torch.equal(v, vid32[k]), k


def test_frame_cnn_pool_argmax_ties_pick_first(mm):
    """Constant frames make every pooling window a tie: the gradient must go to the window's first
    element (torch max_pool2d's rule), so the weight gradients still match the oracle."""
    _lib, ops = mm
    from oracle import mmqg_oracle as O
    B, T, Cin, HW = 2, 3, 3, 40
    frames = torch.ones(B, T, Cin, HW, HW, dtype=torch.float64) * torch.arange(1, T + 1).view(1, T, 1, 1, 1)
    n_frames = torch.full((B,), T)
    vid64 = _cnn_params(Cin, 11)
    leaf = {k: v.clone().requires_grad_("running" not in k) for k, v in vid64.items()}
    want = O.frame_cnn(leaf, frames, n_frames, True)
    g = torch.Generator().manual_seed(5)
    dfeat = torch.randn(want.shape, generator=g, dtype=torch.float64)
    (want * dfeat).sum().backward()
    vid32 = {k: dev(v.float()).requires_grad_("running" not in k) for k, v in vid64.items()}
    got = _cnn_hip(ops, vid32, dev(frames.float()), None, True).reshape(B, T, -1)
    (got * dev(dfeat.float())).sum().backward()
    close(got, want.detach(), tol=5e-4, what="features")
    for k in ("conv4.weight", "conv4.bias", "bn4.weight", "bn4.bias"):
        close(vid32[k].grad, leaf[k].grad, tol=5e-4, what=f"d{k}")


# ---------------------------------------------------------------------------- grouped GEMM
@pytest.mark.parametrize("case", [
    # (a_layout, b_layout, [(M, N, K, beta, ldc_pad)], what)
    (1, 1, [(256, 128, 2048, 1, 0), (256, 300, 1280, 1, 212), (384, 512, 2048, 1, 0)], "one launch, k-slices add atomically"),
    (1, 1, [(2048, 512, 80, 1, 0), (2048, 300, 80, 1, 0)], "one launch, no k split (short K)"),
    (1, 1, [(256, 128, 512, 1, 0), (64, 96, 200, 1, 0)], "a small member: issued one by one"),
    (1, 1, [(256, 256, 512, 0, 0), (256, 256, 512, 1, 0)], "a member that overwrites: one by one"),
    (0, 0, [(130, 257, 300, 1, 3), (256, 128, 64, 1, 0)], "k-major operands: one by one"),
])
def test_grouped_gemm_against_float64(mm, case):
    """mmqg_gemm_f32_grouped: every member C_i = beta*C_i + A_i*B_i vs float64, whichever way it is issued."""
    _lib, ops = mm
    a_layout, b_layout, members, what = case
    g = torch.Generator().manual_seed(len(members) * 1000 + members[0][2])
    probs = (_lib.GemmProblem * len(members))()
    keep, want = [], []
    for i, (M, N, K, beta, pad) in enumerate(members):
        A = torch.randn((M, K) if a_layout == 0 else (K, M), generator=g)
        B = torch.randn((N, K) if b_layout == 0 else (K, N), generator=g)
        C0 = torch.randn(M, N + pad, generator=g)
        A64 = A.double() if a_layout == 0 else A.double().t()
        B64 = B.double().t() if b_layout == 0 else B.double()
        ref = (C0[:, :N].double() if beta else 0) + A64 @ B64
        dA, dB, dC = dev(A), dev(B), dev(C0)
        keep += [dA, dB, dC]
        p = probs[i]
        p.M, p.N, p.K, p.beta = M, N, K, beta
        p.A, p.lda = dA.data_ptr(), A.shape[1]
        p.B, p.ldb = dB.data_ptr(), B.shape[1]
        p.C, p.ldc = dC.data_ptr(), N + pad
        want.append((dC, ref, C0, N, K))
    _lib.check(_lib.load().mmqg_gemm_f32_grouped(a_layout, b_layout, probs, len(members), torch.cuda.current_stream().cuda_stream),
               "gemm_f32_grouped")
    torch.cuda.synchronize()
    for dC, ref, C0, N, K in want:
        close(dC[:, :N], ref, tol=TOL, what=f"grouped gemm ({what}) M{ref.shape[0]} N{N} K{K}")
        assert torch.equal(dC[:, N:].cpu(), C0[:, N:]), f"{what}: wrote past N"


# ------------------------------------------------------------------------------ batch packing
@pytest.mark.parametrize("B,Tf,Tc,Td,inner,Da,Ta", [(5, 4, 7, 6, 40, 12, 4), (3, 3, 5, 4, 3 * 6 * 6, 8, 2), (1, 1, 1, 1, 1, 1, 1)])
def test_pack_batch_bit_exact(mm, B, Tf, Tc, Td, inner, Da, Ta):
    """mmqg_pack_batch: question-major batch -> time-major static inputs, teacher-forcing ids, loss weights,
    audio rows zeroed past n_frames (train.py:149-160,168,175).  Pure data movement: bit-exact."""
    _lib, ops = mm
    g = torch.Generator().manual_seed(B * 100 + Tf)
    frames = torch.randn(B, Tf, inner, generator=g)
    audio = torch.randn(B, Ta, Da, generator=g)
    ctx = torch.randint(0, 50, (B, Tc), generator=g)
    tgt = torch.randint(0, 50, (B, Td), generator=g)
    ctx_len = torch.randint(1, Tc + 1, (B,), generator=g, dtype=torch.int32)
    tgt_len = torch.randint(1, Td + 1, (B,), generator=g, dtype=torch.int32)
    n_frames = torch.randint(0, Tf + 1, (B,), generator=g, dtype=torch.int32)
    stride_b, off = 5 * Da + 7, 3                                   # audio rows live inside a larger per-question block
    d = {k: dev(v) for k, v in dict(frames=frames, audio=audio, ctx=ctx, tgt=tgt, ctx_len=ctx_len, tgt_len=tgt_len,
                                    n_frames=n_frames).items()}
    out = dict(feats=torch.full((Tf, B, inner), 9.0), vals=torch.full((B, stride_b), 9.0), ids_c=torch.full((Tc, B), -1),
               ids_d=torch.full((Td, B), -1), target=torch.full((Td, B), -1), row_w=torch.full((Td, B), 9.0),
               cl=torch.zeros(B, dtype=torch.int32), tl=torch.zeros(B, dtype=torch.int32), nf=torch.zeros(B, dtype=torch.int32))
    o = {k: dev(v) for k, v in out.items()}
    p = _lib.BatchPack(B=B, Tf=Tf, Tc=Tc, Td=Td, Da=Da, audio_rows=Ta, frame_inner=inner, frames=d["frames"].data_ptr(),
                       audio=d["audio"].data_ptr(), context=d["ctx"].data_ptr(), target=d["tgt"].data_ptr(),
                       ctx_len=d["ctx_len"].data_ptr(), tgt_len=d["tgt_len"].data_ptr(), n_frames=d["n_frames"].data_ptr(),
                       start_id=1, feats=o["feats"].data_ptr(), audio_out=o["vals"].data_ptr() + 4 * off,
                       audio_stride_b=stride_b, ids_c=o["ids_c"].data_ptr(), ids_d=o["ids_d"].data_ptr(),
                       target_t=o["target"].data_ptr(), row_w=o["row_w"].data_ptr(), ctx_len_out=o["cl"].data_ptr(),
                       tgt_len_out=o["tl"].data_ptr(), n_frames_out=o["nf"].data_ptr())
    _lib.check(_lib.load().mmqg_pack_batch(C.byref(p), torch.cuda.current_stream().cuda_stream), "pack_batch")
    torch.cuda.synchronize()
    assert torch.equal(o["feats"].cpu(), frames.transpose(0, 1).contiguous())
    keep = (torch.arange(Ta).view(1, -1) < n_frames.view(-1, 1)).unsqueeze(-1)
    vals = o["vals"].cpu()
    assert torch.equal(vals[:, off:off + Ta * Da].view(B, Ta, Da), audio * keep)
    assert bool((vals[:, :off] == 9.0).all()) and bool((vals[:, off + Ta * Da:] == 9.0).all())      # nothing else touched
    assert torch.equal(o["ids_c"].cpu(), ctx.t()) and torch.equal(o["target"].cpu(), tgt.t())
    want_d = torch.cat([torch.ones(1, B, dtype=torch.long), tgt.t()[:-1]], 0)
    assert torch.equal(o["ids_d"].cpu(), want_d)
    want_w = (torch.arange(Td).view(-1, 1) < tgt_len.view(1, -1)).float() / B
    assert torch.equal(o["row_w"].cpu(), want_w)
    assert torch.equal(o["cl"].cpu(), ctx_len) and torch.equal(o["tl"].cpu(), tgt_len) and torch.equal(o["nf"].cpu(), n_frames)
