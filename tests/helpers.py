"""Shared helpers of the parity tests (oracle = checker only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import cdc_oracle as O  # noqa: E402


def make_ids(rng, B, field_dims):
    return np.stack([rng.integers(0, d, size=B) for d in field_dims], axis=1).astype(np.int32)


def sd_cpu(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def oracle_grads(forward_fn, sd, grad_out):
    """Run an oracle forward with autograd on float leaves; returns (out, {name: grad})."""
    leaves = {}
    sd2 = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            t = v.clone().requires_grad_(True)
            leaves[k] = t
            sd2[k] = t
        else:
            sd2[k] = v
    out = forward_fn(sd2)
    out.backward(grad_out)
    return out.detach(), {k: (t.grad if t.grad is not None else None) for k, t in leaves.items()}


def assert_close(a, b, rtol, atol, what=""):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    rec = os.environ.get("CDC_RECORD_MARGINS")
    if rec and a.numel():
        # development aid: how much of the stated tolerance each comparison used (profiles/round3/parity_margins.txt)
        with open(rec, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what}\t{float(err.max()):.3e}\t"
                    f"{float((err / tol).max()):.3e}\t{rtol}\t{atol}\n")
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError(f"{what}: {int(bad.sum())}/{a.numel()} off; worst |d|={err.flatten()[i]:.3e} "
                             f"got {a.flatten()[i]:.6e} want {b.flatten()[i]:.6e} (rtol={rtol}, atol={atol})")


def is_pre_bn_bias(k, names):
    """True for the bias of a Linear whose output feeds a BatchNorm: its gradient is mathematically zero
    (the batch mean removes any constant), so both sides only hold rounding noise there."""
    if k == "shared_bn_bias" or (k.startswith("domain_norm.") and k.endswith(".bias")):
        return "domain_dnns.0.bn.0.running_mean" in names  # STAR: beta_d + beta_s shifts the input of a Linear a BatchNorm follows
    if not k.endswith(".bias"):
        return False
    stem = k[:-5]
    head, _, idx = stem.rpartition(".")
    if ".linears." in k:                                   # DNN: linears.i is normalised by bn.i
        return k.replace(".linears.", ".bn.") in names and "domain_dnn_linears" not in k
    if idx.isdigit() and ".layers." in k:                  # MultiLayerPerceptron: layers.i then layers.i+1 (BatchNorm1d)
        return f"{head}.{int(idx) + 1}.running_mean" in names
    return False


# bf16 gradient bounds (relative L2 per tensor against the exact-accumulation bf16 restatement), see compare_param_grads:
# 5e-2 for a tensor, 2.5e-1 only for the SMALL ones a single flipped row dominates (a gate bias of four elements, one of 30 small
# towers: at most BF16_GRAD_SMALL elements), nine tensors in ten within 5e-2, the median within 2.5e-2
BF16_GRAD_MAX, BF16_GRAD_MAX_SMALL, BF16_GRAD_SMALL, BF16_GRAD_P90, BF16_GRAD_MEDIAN = 5e-2, 2.5e-1, 2048, 5e-2, 2.5e-2


def compare_param_grads(named_params, want, rtol, atol, bf16=False, all_names=None, bn_active=True, max_rel=None, median_rel=None,
                        p90_rel=None):
    """Gradient check per parameter (pre-BatchNorm biases only have to be negligible)."""
    names = set(all_names) if all_names is not None else set(want)
    rels = []
    for k, g in want.items():
        if g is None:
            continue
        got = named_params[k].grad
        assert got is not None, f"no gradient for {k}"
        scale = max(float(g.abs().max()), 1e-6)
        if bn_active and is_pre_bn_bias(k, names):
            wk = {"shared_bn_bias": "shared_bn_weight"}.get(k, k[:-5] + ".weight")
            wscale = float(want[wk].abs().max())
            # fp32: summation noise of a mathematically zero sum.  bf16: the bias gradient is the column sum of the bf16-rounded
            # dZ the weight gradient contracts (csrc/gemm2.hip), so the zero sum carries the roundings' noise (~2^-9 per element)
            bound = (1e-1 if bf16 else 1e-3) * max(wscale, 1e-3) + 1e-4
            assert float(got.abs().max()) <= bound, f"pre-BN bias grad {k} not ~0: {float(got.abs().max()):.3e} > {bound:.3e}"
            continue
        if bf16:
            # Against the oracle's bf16 restatement with exact accumulation (O.MATMUL_BF16 = "exact"): relative L2 error per gradient
            # tensor.  What the bound has to leave room for is not rounding but BRANCHES: an activation that lands on the other side
            # of a bf16 rounding boundary moves the next layer's pre-activations by ~2.5e-4 sigma, a relu unit that close to zero takes
            # the other branch, and that row's whole upstream gradient moves.  The density of such flips does not fall with the
            # batch size, so neither does the error.  Measured (tools/grad_err_report.py, profiles/round2/grad_err.txt), max / median
            # over the tensors of a model:
            #     two CPU restatements, accumulation fp32 vs exact:  MMoE-8 1.7e-2 / 7.0e-3   PLE-3 1.6e-1 / 1.1e-2   STAR-30 6.8e-2 / 5.6e-4
            #     HIP bf16 vs the exact restatement:                 MMoE-8 1.8e-2 / 8.5e-3   PLE-3 8.5e-3 / 3.3e-3   STAR-30 1.1e-1 / 4.3e-3
            # (vs the fp32 oracle 1.8e-1 / 1.2e-1: bf16 operand rounding flips ~1e-3 of the units).  So: the MEDIAN over tensors — what a
            # systematic error (a missed term, a wrong rounding point) moves — is held to 3x the worst measured median, nine tensors in
            # ten to 5e-2, a single tensor to 5e-2 as well unless it is SMALL (a gate bias of four elements downstream of a flipped row;
            # one of 30 small towers: <= 2048 elements), which may reach 2.5e-1.  A model small enough to have no flips is held to
            # 5e-4 in tests/test_gpu_gaps.py, PLE-3's rows without a flip to 5e-4 in tests/test_gpu_cgc_mid.py.
            gd, wd = got.detach().cpu().double(), g.double()
            rel = float((gd - wd).norm() / max(float(wd.norm()), 1e-12))
            rels.append(rel)
            bound = max_rel or (BF16_GRAD_MAX_SMALL if gd.numel() <= BF16_GRAD_SMALL else BF16_GRAD_MAX)
            assert rel < bound or float((gd - wd).abs().max()) < 1e-5, f"grad {k} ({gd.numel()} elements): relative L2 error {rel:.3e} vs bf16 restatement (bound {bound})"
        else:
            assert_close(got, g, rtol, atol * max(scale, 1.0), f"grad {k}")
    if bf16 and len(rels) >= 8:
        med, p90 = float(np.median(rels)), float(np.quantile(rels, 0.9))
        assert med < (median_rel or BF16_GRAD_MEDIAN), f"median relative L2 error of the gradient tensors {med:.3e}"
        assert p90 < (p90_rel or BF16_GRAD_P90), f"90th percentile of the gradient tensors' relative L2 errors {p90:.3e}"
