"""PLE on the HIP path vs the CPU oracle (forward, every parameter gradient, BatchNorm statistics)."""
import numpy as np
import pytest
import torch

from helpers import O, assert_close, compare_param_grads, make_ids, oracle_grads, sd_cpu

pytestmark = pytest.mark.gpu

# fp32 MFMA path: a k-ordered fmaf chain; tolerance covers summation-order differences vs MKL only
F32_RTOL, F32_ATOL = 2e-4, 2e-5
# bf16 operands (8 significant bits) with fp32 accumulation
BF16_RTOL, BF16_ATOL = 5e-2, 5e-3


def _build(cuda, field_dims, precision, B, seed=0, dims=((32, 16), (8,)), tower=(8, 4), D=4, n_tower=3):
    from cdcmdr_amd.model.ple import PLE
    torch.manual_seed(seed)
    m = PLE(field_dims, D, n_tower, 2, 2, dims, tower, dropout=0.0).to(cuda)
    m.set_precision(precision)
    rng = np.random.default_rng(seed + 1)
    x = make_ids(rng, B, field_dims)
    return m, x


@pytest.mark.parametrize("precision,B", [("f32", 64), ("f32", 1), ("f32", 257), ("bf16", 64)])
def test_ple_forward_backward(cuda, precision, B):
    field_dims = [7, 100, 3, 50, 11, 29]
    m, x = _build(cuda, field_dims, precision, B)
    m.train()
    sd = sd_cpu(m)
    xg = torch.from_numpy(x).to(cuda)
    out = m(xg)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
    out.backward(gout.to(cuda))
    stats = {}
    ref, grads = oracle_grads(lambda s: O.ple_forward(s, x, field_dims, 3, training=True, stats_out=stats), sd, gout)
    rtol, atol = (F32_RTOL, F32_ATOL) if precision == "f32" else (BF16_RTOL, BF16_ATOL)
    assert_close(out, ref, rtol, atol, "probabilities")
    compare_param_grads(dict(m.named_parameters()), grads, rtol, atol, bf16=(precision == "bf16"), all_names=list(sd), bn_active=(B > 1))
    new_sd = sd_cpu(m)
    for k, v in stats.items():
        assert_close(new_sd[k], v, rtol, atol, f"stat {k}")
