"""The two layers of a level's experts as one forward launch (csrc/pair.hip, cdc_expert_pair_fwd) against the two grouped
launches it replaces (cdc_gemm_bf16_nt) — same operands, rounding points, accumulation order and dropout streams, so every
output is held to BIT equality: through the C-ABI on raw operands (ragged row counts, riders, dropout on / off) and through the
PLE model (forward, and the gradients that the unchanged backward launches form from the fused forward's outputs)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from helpers import make_ids

pytestmark = pytest.mark.gpu

H1, H2 = 256, 128


def _err(lib):
    msg = lib.cdc_last_error()
    return msg.decode() if msg else ""


def _unfused(L, lib, dev, M, K, x, w1, b1, w2, b2, ws, bs, relu, drop_p, seed1, seed2, step_dev):
    """the two cdc_gemm_bf16_nt forward launches of plan.GLinear._build_fwd_g2: layer 1 (hidden as bf16 only), layer 2 + riders"""
    n, Kr = len(w1), x.shape[1]
    h = torch.zeros((M, n * H1), dtype=torch.bfloat16, device=dev)
    y = torch.zeros((M, n * H2), dtype=torch.float32, device=dev)
    ys = [torch.zeros((M, w.shape[0]), dtype=torch.float32, device=dev) for w in ws]
    s = torch.cuda.current_stream().cuda_stream
    a = L.G2Args()
    a.n_out = a.n_seg = n
    a.mode, a.relu, a.drop_p, a.mask_scale, a.seed, a.seed_offset_dev = 0, relu, drop_p, 1.0, seed1, step_dev.data_ptr()
    for i in range(n):
        O, S = a.o[i], a.s[i]
        O.y, O.yh, O.ldyh = None, h.data_ptr() + 2 * i * H1, h.stride(0)
        O.bias, O.mask, O.bn_partial = b1[i].data_ptr(), None, None
        O.M, O.N, O.act_cols, O.accumulate, O.mask_bf16, O.stream_id = M, H1, H1, 0, 0, i
        S.a, S.lda, S.b, S.ldb, S.Kr, S.out = x.data_ptr(), x.stride(0), w1[i].data_ptr(), w1[i].stride(0), Kr, i
    assert lib.cdc_gemm_bf16_nt(C.byref(a), s) == 0, _err(lib)
    b = L.G2Args()
    b.n_out = b.n_seg = n + len(ws)
    b.mode, b.relu, b.drop_p, b.mask_scale, b.seed, b.seed_offset_dev = 0, relu, drop_p, 1.0, seed2, step_dev.data_ptr()
    for i in range(n):
        O, S = b.o[i], b.s[i]
        O.y, O.ldy, O.yh = y.data_ptr() + 4 * i * H2, y.stride(0), None
        O.bias, O.mask, O.bn_partial = b2[i].data_ptr(), None, None
        O.M, O.N, O.act_cols, O.accumulate, O.mask_bf16, O.stream_id = M, H2, H2, 0, 0, i
        S.a, S.lda, S.b, S.ldb, S.Kr, S.out = h.data_ptr() + 2 * i * H1, h.stride(0), w2[i].data_ptr(), w2[i].stride(0), H1, i
    for j, w in enumerate(ws):
        O, S = b.o[n + j], b.s[n + j]
        O.y, O.ldy, O.yh = ys[j].data_ptr(), ys[j].stride(0), None
        O.bias, O.mask, O.bn_partial = bs[j].data_ptr(), None, None
        O.M, O.N, O.act_cols, O.accumulate, O.mask_bf16, O.stream_id = M, w.shape[0], 0, 0, 0, n + j
        S.a, S.lda, S.b, S.ldb, S.Kr, S.out = x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), Kr, n + j
    assert lib.cdc_gemm_bf16_nt(C.byref(b), s) == 0, _err(lib)
    torch.cuda.synchronize()
    return h, y, ys


def _fused(L, lib, dev, M, K, x, w1, b1, w2, b2, ws, bs, relu, drop_p, seed1, seed2, step_dev):
    n, Kr = len(w1), x.shape[1]
    h = torch.zeros((M, n * H1), dtype=torch.bfloat16, device=dev)
    y = torch.zeros((M, n * H2), dtype=torch.float32, device=dev)
    yh = torch.zeros((M, n * H2), dtype=torch.bfloat16, device=dev)
    ys = [torch.zeros((M, w.shape[0]), dtype=torch.float32, device=dev) for w in ws]
    a = L.ExpertPairArgs()
    a.n_expert, a.M, a.K1r, a.H1, a.H2, a.relu, a.drop_p = n, M, Kr, H1, H2, relu, drop_p
    a.seed1, a.seed2, a.seed_offset_dev = seed1, seed2, step_dev.data_ptr()
    for i in range(n):
        E = a.e[i]
        E.x, E.ldx = x.data_ptr(), x.stride(0)
        E.w1, E.ldw1, E.b1 = w1[i].data_ptr(), w1[i].stride(0), b1[i].data_ptr()
        E.w2, E.ldw2, E.b2 = w2[i].data_ptr(), w2[i].stride(0), b2[i].data_ptr()
        E.h, E.ldh = h.data_ptr() + 2 * i * H1, h.stride(0)
        E.y, E.ldy = y.data_ptr() + 4 * i * H2, y.stride(0)
        E.yh, E.ldyh = yh.data_ptr() + 2 * i * H2, yh.stride(0)
        E.stream1 = E.stream2 = i
        E.ws, E.ns = None, 0
    for j, w in enumerate(ws):
        E = a.e[j]
        E.ws, E.ldws, E.bs, E.ys, E.ldys, E.ns = w.data_ptr(), w.stride(0), bs[j].data_ptr(), ys[j].data_ptr(), ys[j].stride(0), w.shape[0]
    assert lib.cdc_expert_pair_fwd(C.byref(a), torch.cuda.current_stream().cuda_stream) == 0, _err(lib)
    torch.cuda.synchronize()
    return h, y, yh, ys


@pytest.mark.parametrize("M,K,n,riders,drop_p", [(4096, 416, 8, (4, 4, 4, 8), 0.2), (300, 416, 3, (4, 16), 0.2), (128, 64, 1, (), 0.0),
                                                  (1, 100, 2, (3,), 0.5), (1000, 832, 4, (4, 4, 4, 4), 0.0)])
def test_expert_pair_bit_equals_two_grouped_launches(cuda, M, K, n, riders, drop_p):
    from cdcmdr_amd import _lib as L
    lib = L.load()
    dev = cuda
    g = torch.Generator().manual_seed(M * 7 + K)
    Kr = (K + 63) // 64 * 64

    def bf(rows, cols, pad, scale):
        t = torch.zeros((rows, pad), dtype=torch.bfloat16)
        t[:, :cols] = (torch.randn((rows, cols), generator=g) * scale).to(torch.bfloat16)
        return t.to(dev)
    x = bf(M, K, Kr, 1.0)
    w1 = [bf(H1, K, Kr, K ** -0.5) for _ in range(n)]
    w2 = [bf(H2, H1, H1, H1 ** -0.5) for _ in range(n)]
    b1 = [(torch.randn(H1, generator=g) * 0.1).to(dev) for _ in range(n)]
    b2 = [(torch.randn(H2, generator=g) * 0.1).to(dev) for _ in range(n)]
    ws = [bf(r, K, Kr, K ** -0.5) for r in riders]
    bs = [(torch.randn(r, generator=g) * 0.1).to(dev) for r in riders]
    step_dev = torch.full((1,), 5, dtype=torch.int32, device=dev)
    args = (L, lib, dev, M, K, x, w1, b1, w2, b2, ws, bs, 1, drop_p, 0x1234ABCD5678, 0x9999AAAA1111, step_dev)
    h_u, y_u, ys_u = _unfused(*args)
    h_f, y_f, yh_f, ys_f = _fused(*args)
    assert torch.equal(h_u.view(torch.int16), h_f.view(torch.int16)), "hidden activation differs"
    assert torch.equal(y_u, y_f), f"layer-2 output differs: max |d| {float((y_u - y_f).abs().max()):.3e}"
    assert torch.equal(yh_f.view(torch.int16), y_f.to(torch.bfloat16).view(torch.int16))
    for j in range(len(ws)):
        assert torch.equal(ys_u[j], ys_f[j]), f"rider {j} differs"
    if drop_p > 0:                                    # the dropout is alive in both layers
        assert 0.5 * drop_p < float((h_f.float() == 0).float().mean()) and float((y_f == 0).float().mean()) > 0.5 * drop_p
    assert float(y_f.abs().max()) > 0


def test_expert_pair_rejects_other_widths_and_bad_arguments(cuda):
    from cdcmdr_amd import _lib as L
    lib = L.load()
    a = L.ExpertPairArgs()
    a.n_expert, a.M, a.K1r, a.H1, a.H2 = 1, 128, 64, 128, 64
    assert lib.cdc_expert_pair_fwd(C.byref(a), None) != 0 and "256, 128" in _err(lib)
    a.H1, a.H2, a.K1r = 256, 128, 100
    assert lib.cdc_expert_pair_fwd(C.byref(a), None) != 0 and "multiple of 64" in _err(lib)
    a.K1r, a.n_expert = 64, 17
    assert lib.cdc_expert_pair_fwd(C.byref(a), None) != 0
    a.n_expert = 1
    assert lib.cdc_expert_pair_fwd(C.byref(a), None) != 0 and "malformed" in _err(lib)      # null operands


FD = [1000] * 26
DIMS, TOWER, D = ((256, 128), (64,)), (64, 32), 16


@pytest.fixture
def _pair_env():
    from cdcmdr_amd import plan as P
    yield
    P.ExpertPair.enabled = True


@pytest.mark.parametrize("n_tower,B,dropout,train", [(3, 4096, 0.2, True), (3, 200, 0.0, True), (4, 1000, 0.2, True), (3, 300, 0.2, False)])
def test_ple_with_the_fused_expert_pair_equals_the_two_launches(cuda, _pair_env, n_tower, B, dropout, train):
    from cdcmdr_amd import plan as P
    from cdcmdr_amd.model.ple import PLE
    rng = np.random.default_rng(B)
    x = torch.from_numpy(make_ids(rng, B, FD)).to(cuda)
    gout = torch.randn((B, n_tower), generator=torch.Generator().manual_seed(7)).to(cuda)
    res = {}
    for fused in (False, True):
        P.ExpertPair.enabled = fused
        torch.manual_seed(0)
        m = PLE(FD, D, n_tower, 2, 2, DIMS, TOWER, dropout=dropout).to(cuda).set_precision("bf16")
        m.seed = 1234
        m.train(train)
        if train:
            out = m(x)
            out.backward(gout)
            grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        else:
            with torch.no_grad():
                out = m(x)
            grads = {}
        assert any(isinstance(op, P.ExpertPair) for op in m.plan_holder(B).plan.ops) == fused
        res[fused] = (out.detach().clone(), grads)
    assert torch.equal(res[False][0], res[True][0]), "forward differs"
    assert set(res[False][1]) == set(res[True][1])
    for k in res[False][1]:
        assert torch.equal(res[False][1][k], res[True][1][k]), f"gradient of {k} differs"
