"""PLE on the HIP path vs the CPU oracle (forward, every parameter gradient, BatchNorm statistics)."""
import numpy as np
import pytest
import torch

from helpers import O, assert_close, compare_param_grads, make_ids, oracle_grads, sd_cpu

pytestmark = pytest.mark.gpu

# fp32 MFMA path: a k-ordered fmaf chain; tolerance covers summation-order differences vs MKL only
F32_RTOL, F32_ATOL = 2e-4, 2e-5
# bf16 path vs the oracle's bf16 restatement: identical rounded operands, only the fp32 accumulation order differs
BF16_RTOL, BF16_ATOL = 5e-3, 2e-3   # one relu flip near zero moves a probability by ~1e-3


def _build(cuda, field_dims, precision, B, seed=0, dims=((32, 16), (8,)), tower=(8, 4), D=4, n_tower=3):
    if precision == "bf16":                     # realistic widths: bf16 rounding averages out over K = 416 / 256 / 128
        dims, tower, D = ((256, 128), (64,)), (64, 32), 16
    from cdcmdr_amd.model.ple import PLE
    torch.manual_seed(seed)
    m = PLE(field_dims, D, n_tower, 2, 2, dims, tower, dropout=0.0).to(cuda)
    m.set_precision(precision)
    rng = np.random.default_rng(seed + 1)
    x = make_ids(rng, B, field_dims)
    return m, x


@pytest.mark.parametrize("precision,B", [("f32", 64), ("f32", 1), ("f32", 257), ("bf16", 512)])
def test_ple_forward_backward(cuda, precision, B):
    field_dims = [7, 100, 3, 50, 11, 29] if precision == "f32" else [1000] * 26
    m, x = _build(cuda, field_dims, precision, B)
    m.train()
    sd = sd_cpu(m)
    xg = torch.from_numpy(x).to(cuda)
    out = m(xg)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
    out.backward(gout.to(cuda))
    stats = {}
    # bf16 path: the oracle restates the SAME arithmetic (operands of every contraction rounded to bf16, fp32 accumulate),
    # so the comparison stays at accumulation-order tolerance instead of a loose "bf16 noise" bound
    O.MATMUL_BF16 = "exact" if precision == "bf16" else False
    try:
        ref, grads = oracle_grads(lambda s: O.ple_forward(s, x, field_dims, 3, training=True, stats_out=stats), sd, gout)
    finally:
        O.MATMUL_BF16 = False
    rtol, atol = (F32_RTOL, F32_ATOL) if precision == "f32" else (BF16_RTOL, BF16_ATOL)
    assert_close(out, ref, rtol, atol, "probabilities")
    compare_param_grads(dict(m.named_parameters()), grads, rtol, atol, bf16=(precision == "bf16"), all_names=list(sd), bn_active=(B > 1))
    new_sd = sd_cpu(m)
    for k, v in stats.items():
        assert_close(new_sd[k], v, rtol, atol, f"stat {k}")
