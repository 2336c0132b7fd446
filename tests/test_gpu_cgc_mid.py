"""The fused PLE level boundary (csrc/cgc.hip: pooling -> next level's experts + gates -> pooling in one launch per
direction) against the three separate launches it replaces (cdc_gate_pool_* / cdc_gemm_bf16_nt), and against the oracle.

The forward keeps every rounding point, summation order and the dropout stream of the unfused chain, so its results are held
to BIT equality; the backward sums the gate-gradient dot products with 16 lanes per row instead of H/4, so it is held to a
rounding-level bound on a model without dropout and to the oracle's bf16 restatement like every other bf16 path."""
import os

import numpy as np
import pytest
import torch

from helpers import O, assert_close, compare_param_grads, make_ids, oracle_grads, sd_cpu

pytestmark = pytest.mark.gpu

FD = [1000] * 26
DIMS, TOWER, D = ((256, 128), (64,)), (64, 32), 16


def _model(cuda, n_tower, dropout, seed=0, fused=True):
    from cdcmdr_amd.model.ple import PLE
    torch.manual_seed(seed)
    m = PLE(FD, D, n_tower, 2, 2, DIMS, TOWER, dropout=dropout).to(cuda).set_precision("bf16")
    m.seed = 1234
    os.environ["CDC_CGC_MID"] = "1" if fused else "0"
    return m


def _uses_fused(m, B):
    from cdcmdr_amd import plan as P
    holder = m.plan_holder(B)
    return any(isinstance(op, P.CGCMid) for op in holder.plan.ops)


@pytest.fixture(autouse=True)
def _restore_env():
    old = os.environ.get("CDC_CGC_MID")
    yield
    if old is None:
        os.environ.pop("CDC_CGC_MID", None)
    else:
        os.environ["CDC_CGC_MID"] = old


@pytest.mark.parametrize("n_tower,B,dropout", [(3, 4096, 0.2), (3, 100, 0.0), (4, 1000, 0.2), (3, 16, 0.0), (3, 1, 0.0)])
def test_fused_boundary_equals_the_three_launches(cuda, n_tower, B, dropout):
    rng = np.random.default_rng(B)
    x = torch.from_numpy(make_ids(rng, B, FD)).to(cuda)
    gout = torch.randn((B, n_tower), generator=torch.Generator().manual_seed(7)).to(cuda)
    res = {}
    for fused in (False, True):
        m = _model(cuda, n_tower, dropout, fused=fused)
        m.train()
        out = m(x)
        assert _uses_fused(m, B) == fused
        out.backward(gout)
        res[fused] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    out_u, g_u = res[False]
    out_f, g_f = res[True]
    assert torch.equal(out_u, out_f), f"forward differs: max |d| {float((out_u - out_f).abs().max()):.3e}"
    assert set(g_u) == set(g_f)
    worst = 0.0
    for k in g_u:
        a, b = g_u[k].double(), g_f[k].double()
        rel = float((a - b).norm() / max(float(a.norm()), 1e-30))
        worst = max(worst, rel)
        # same operands and rounding points; only the lane count of the row dot products differs, which can move a bf16 rounding
        # of a gate-logit gradient by one unit here and there
        assert rel < 1e-4, f"{k}: fused vs unfused relative L2 {rel:.3e}"
    print(f"n_tower {n_tower} B {B}: forward bit-equal; worst gradient relative L2 fused vs unfused {worst:.2e}")


def test_fused_boundary_against_the_oracle(cuda):
    B = 512
    m = _model(cuda, 3, 0.0, fused=True)
    m.train()
    rng = np.random.default_rng(3)
    x = make_ids(rng, B, FD)
    sd = sd_cpu(m)
    out = m(torch.from_numpy(x).to(cuda))
    assert _uses_fused(m, B)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
    out.backward(gout.to(cuda))
    O.MATMUL_BF16 = "exact"
    try:
        ref, grads = oracle_grads(lambda s: O.ple_forward(s, x, FD, 3, training=True, stats_out={}), sd, gout)
    finally:
        O.MATMUL_BF16 = False
    assert_close(out, ref, 5e-3, 2e-3, "probabilities")
    compare_param_grads(dict(m.named_parameters()), grads, 5e-3, 2e-3, bf16=True, all_names=list(sd))


def test_fused_boundary_eval_mode(cuda):
    B = 300
    rng = np.random.default_rng(11)
    x = torch.from_numpy(make_ids(rng, B, FD)).to(cuda)
    outs = {}
    for fused in (False, True):
        m = _model(cuda, 3, 0.2, fused=fused)
        m.eval()
        with torch.no_grad():
            outs[fused] = m(x).clone()
    assert torch.equal(outs[False], outs[True])
