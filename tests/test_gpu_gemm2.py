"""The bf16-operands-from-memory contraction path (csrc/gemm2.hip) through the C-ABI.

Checked (a) against the same arithmetic restated on the CPU in double on the bf16-rounded operands, and (b) bit for bit against
the register-staged bf16 path of csrc/gemm.hip (Plan(g2=False): what ragged launches take), which rounds the same fp32 values to the same bf16 operands and
accumulates the same K-slabs in the same order."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _r64(n):
    return (n + 63) // 64 * 64


def _shadow(t, pad_rows=True):
    """zero-padded bf16 copy [rows64, cols64 + 64] of an fp32 matrix, as plan.py allocates shadows"""
    rows, cols = t.shape
    s = torch.zeros((_r64(rows) if pad_rows else rows, _r64(cols) + 64), dtype=torch.bfloat16, device=t.device)
    s[:rows, :cols] = t.to(torch.bfloat16)
    return s


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("M,N,K", [(4096, 256, 416), (300, 20, 416), (128, 64, 64), (77, 130, 100), (1, 5, 8), (513, 129, 65)])
def test_forward_against_the_restated_arithmetic(cuda, M, N, K):
    from cdcmdr_amd import _lib as L
    lib = L.load()
    gen = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=gen).to(cuda)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(cuda)
    b = torch.randn(N, generator=gen).to(cuda)
    xh, wh = _shadow(x), _shadow(w, pad_rows=False)
    y = torch.full((M, N + 3), float("nan"), device=cuda)                 # odd row stride: the scalar store path
    yh = torch.zeros((M, _r64(N)), dtype=torch.bfloat16, device=cuda)
    a = L.G2Args()
    a.n_out = a.n_seg = 1
    a.mode, a.relu, a.drop_p, a.mask_scale = 0, 1, 0.0, 1.0
    O, S = a.o[0], a.s[0]
    O.y, O.ldy, O.yh, O.ldyh, O.bias = y.data_ptr(), y.stride(0), yh.data_ptr(), yh.stride(0), b.data_ptr()
    O.M, O.N, O.act_cols = M, N, N // 2                                   # relu on the first half of the columns only
    S.a, S.lda, S.b, S.ldb, S.Kr, S.out = xh.data_ptr(), xh.stride(0), wh.data_ptr(), wh.stride(0), _r64(K), 0
    L.check(lib.cdc_gemm_bf16_nt(C.byref(a), _stream()), "gemm_bf16_nt")
    want = xh[:M, :K].double().cpu() @ wh[:N, :K].double().cpu().t() + b.double().cpu()
    want[:, :N // 2] = torch.relu(want[:, :N // 2])
    assert_close(y[:, :N], want, 2e-5, 2e-5, "y")
    assert bool(torch.isnan(y[:, N:]).all()), "wrote past the N columns"
    assert torch.equal(yh[:, :N].cpu(), y[:, :N].to(torch.bfloat16).cpu()), "the bf16 shadow is not the rounding of the fp32 output"
    assert bool((yh[:, N:] == 0).all())


def test_grad_input_segments_mask_and_accumulate(cuda):
    """mode 1: three segments reduce into one output, the producing layer's activation mask is applied, the result is added to
    what the fp32 destination holds, and the shadow carries the accumulated value."""
    from cdcmdr_amd import _lib as L
    lib = L.load()
    gen = torch.Generator().manual_seed(3)
    M, K, Ns = 260, 96, [40, 64, 7]
    dzs = [torch.randn(M, n, generator=gen).to(cuda) for n in Ns]
    ws = [(torch.randn(n, K, generator=gen) / n ** 0.5).to(cuda) for n in Ns]
    y_prev = torch.randn(M, K, generator=gen).to(cuda)                    # activation of the producing layer (mask source)
    old = torch.randn(M, K, generator=gen).to(cuda)
    dx = old.clone()
    dxh = torch.zeros((M, _r64(K)), dtype=torch.bfloat16, device=cuda)
    keep = []
    a = L.G2Args()
    a.n_out, a.n_seg = 1, len(Ns)
    a.mode, a.relu, a.drop_p, a.mask_scale = 1, 0, 0.0, 1.25
    O = a.o[0]
    O.y, O.ldy, O.yh, O.ldyh = dx.data_ptr(), dx.stride(0), dxh.data_ptr(), dxh.stride(0)
    O.mask, O.ldmask, O.mask_bf16 = y_prev.data_ptr(), y_prev.stride(0), 0
    O.M, O.N, O.act_cols, O.accumulate = M, K, K - 10, 1                  # the last 10 columns are not masked
    want = torch.zeros(M, K, dtype=torch.float64)
    for i, (dz, w) in enumerate(zip(dzs, ws)):
        dzh, wth = _shadow(dz), _shadow(w.t().contiguous(), pad_rows=False)
        keep += [dzh, wth]
        S = a.s[i]
        S.a, S.lda, S.b, S.ldb, S.Kr, S.out = dzh.data_ptr(), dzh.stride(0), wth.data_ptr(), wth.stride(0), _r64(Ns[i]), 0
        want += dzh[:M, :Ns[i]].double().cpu() @ wth[:K, :Ns[i]].double().cpu().t()
    L.check(lib.cdc_gemm_bf16_nt(C.byref(a), _stream()), "gemm_bf16_nt")
    mk = (y_prev[:, :K - 10] > 0).double().cpu()
    want[:, :K - 10] = want[:, :K - 10] * 1.25 * mk
    want = want + old.double().cpu()
    assert_close(dx, want, 2e-5, 2e-5, "dx")
    assert torch.equal(dxh[:, :K].cpu(), dx.to(torch.bfloat16).cpu())


def _run_plan(cuda, monkeypatch, g2, M, K, Ns, seed):
    from cdcmdr_amd import plan as P
    plan = P.Plan(cuda, M, precision="bf16", training=True, dropout=0.0, g2=g2)
    assert plan.use_g2 == g2
    gen = torch.Generator().manual_seed(seed)
    xb = plan.new(K)
    xb.tensor().copy_(torch.randn(M, K, generator=gen))
    ws = [torch.nn.Parameter((torch.randn(n, K, generator=gen) / K ** 0.5).to(cuda)) for n in Ns]
    bs = [torch.nn.Parameter(torch.randn(n, generator=gen).to(cuda)) for n in Ns]
    l1 = P.GLinear(plan, [{"x": xb, "w": w, "b": b} for w, b in zip(ws, bs)], relu=True)
    # a second layer on the first output: its grad-input applies the first layer's activation mask and (g2) hands the
    # gradient shadow straight to the first layer's backward
    w2 = torch.nn.Parameter((torch.randn(24, Ns[0], generator=gen) / Ns[0] ** 0.5).to(cuda))
    l2 = P.GLinear(plan, [{"x": l1.outs[0], "w": w2, "b": None}])
    outs = [l2.outs[0]] + l1.outs[1:]
    plan.finalize(outs)
    plan.forward()
    for o in outs:
        o.grad.tensor().copy_(torch.randn(o.rows, o.cols, generator=gen))
    for o in l1.outs[1:]:                                     # relu-fused outputs hold dZ: apply their mask by hand
        o.grad.tensor().mul_((o.tensor() > 0).float())
    plan.backward()
    res = {"y%d" % i: o.tensor().clone() for i, o in enumerate(outs)}
    res["dx"] = xb.grad.tensor().clone()
    for i, p in enumerate(ws + bs + [w2]):
        res["dp%d" % i] = plan.param_grads[id(p)].clone()
    return res


@pytest.mark.parametrize("M,K,Ns", [(4096, 416, [256, 8, 4]), (200, 100, [72, 16]), (64, 64, [64])])
def test_bits_equal_the_register_staged_bf16_path(cuda, monkeypatch, M, K, Ns):
    new = _run_plan(cuda, monkeypatch, True, M, K, Ns, seed=11)
    old = _run_plan(cuda, monkeypatch, False, M, K, Ns, seed=11)
    assert new.keys() == old.keys()
    n_w = len(Ns)
    for k in new:
        if k.startswith("dp") and n_w <= int(k[2:]) < 2 * n_w:
            # bias gradients: column sums of the bf16 dZ shadow (one MFMA against ones) vs of the fp32 dZ
            assert_close(new[k], old[k], 2e-2, 2e-3 * float(old[k].abs().max()), k)
            continue
        if k.startswith("dp"):
            # weight gradients: the same products, but the batch rows are cut into a different number of slices (larger output
            # tiles -> fewer of them -> other split) whose fp32 partial sums are added in another grouping
            assert_close(new[k], old[k], 2e-5, 1e-5 * float(old[k].abs().max()), k)
            continue
        assert torch.equal(new[k], old[k]), f"{k}: max |d| {float((new[k] - old[k]).abs().max()):.3e}"


def test_weight_and_activation_shadows(cuda):
    from cdcmdr_amd import _lib as L
    lib = L.load()
    gen = torch.Generator().manual_seed(5)
    ws = [torch.randn(n, k, generator=gen).to(cuda) for n, k in [(37, 100), (256, 416), (4, 64)]]
    a = L.WShadowArgs()
    a.n = len(ws)
    hs, ts = [], []
    for i, w in enumerate(ws):
        N, K = w.shape
        h = torch.zeros((N, _r64(K)), dtype=torch.bfloat16, device=cuda)
        t = torch.zeros((K, _r64(N)), dtype=torch.bfloat16, device=cuda)
        hs.append(h), ts.append(t)
        T = a.t[i]
        T.src, T.dst_h, T.ld_h, T.dst_t, T.ld_t, T.rows, T.cols = w.data_ptr(), h.data_ptr(), h.stride(0), t.data_ptr(), t.stride(0), N, K
    L.check(lib.cdc_weight_shadows(C.byref(a), _stream()), "weight_shadows")
    for w, h, t in zip(ws, hs, ts):
        N, K = w.shape
        assert torch.equal(h[:, :K], w.to(torch.bfloat16)) and bool((h[:, K:] == 0).all())
        assert torch.equal(t[:, :N], w.t().to(torch.bfloat16)) and bool((t[:, N:] == 0).all())
    x = torch.randn(301, 77, generator=gen).to(cuda)
    big = torch.zeros((301, 200), device=cuda)
    big[:, 8:85] = x                                                       # a column slice with a row stride
    d = torch.zeros((320, 256), dtype=torch.bfloat16, device=cuda)
    s = L.ShadowArgs()
    s.n = 1
    s.t[0].src, s.t[0].ld_src = big.data_ptr() + 4 * 8, big.stride(0)
    s.t[0].dst, s.t[0].ld_dst = d.data_ptr() + 2 * 8, d.stride(0)
    s.t[0].rows, s.t[0].cols = 301, 77
    L.check(lib.cdc_shadow_bf16(C.byref(s), _stream()), "shadow_bf16")
    assert torch.equal(d[:301, 8:85], x.to(torch.bfloat16))
    d[:301, 8:85] = 0
    assert bool((d == 0).all()), "the conversion wrote outside its view"
