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
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError(f"{what}: {int(bad.sum())}/{a.numel()} off; worst |d|={err.flatten()[i]:.3e} "
                             f"got {a.flatten()[i]:.6e} want {b.flatten()[i]:.6e} (rtol={rtol}, atol={atol})")


def is_pre_bn_bias(k, names):
    """True for the bias of a Linear whose output feeds a BatchNorm: its gradient is mathematically zero
    (the batch mean removes any constant), so both sides only hold rounding noise there."""
    if not k.endswith(".bias"):
        return False
    stem = k[:-5]
    head, _, idx = stem.rpartition(".")
    if ".linears." in k:                                   # DNN: linears.i is normalised by bn.i
        return k.replace(".linears.", ".bn.") in names and "domain_dnn_linears" not in k
    if idx.isdigit() and ".layers." in k:                  # MultiLayerPerceptron: layers.i then layers.i+1 (BatchNorm1d)
        return f"{head}.{int(idx) + 1}.running_mean" in names
    return False


def compare_param_grads(named_params, want, rtol, atol, bf16=False, all_names=None, bn_active=True):
    """Gradient check per parameter (pre-BatchNorm biases only have to be negligible)."""
    names = set(all_names) if all_names is not None else set(want)
    for k, g in want.items():
        if g is None:
            continue
        got = named_params[k].grad
        assert got is not None, f"no gradient for {k}"
        scale = max(float(g.abs().max()), 1e-6)
        if bn_active and is_pre_bn_bias(k, names):
            wscale = float(want[k[:-5] + ".weight"].abs().max())
            # fp32: summation noise of a mathematically zero sum.  bf16: the bias gradient is the column sum of the bf16-rounded
            # dZ the weight gradient contracts (csrc/gemm2.hip), so the zero sum carries the roundings' noise (~2^-9 per element)
            bound = (1e-1 if bf16 else 1e-3) * max(wscale, 1e-3) + 1e-4
            assert float(got.abs().max()) <= bound, f"pre-BN bias grad {k} not ~0: {float(got.abs().max()):.3e} > {bound:.3e}"
            continue
        if bf16:
            # against the oracle's bf16 restatement; a relu unit whose pre-activation sits within fp32 accumulation noise of
            # zero may still flip, so the bound is on the relative L2 error of each gradient tensor
            gd, wd = got.detach().cpu().double(), g.double()
            rel = float((gd - wd).norm() / max(float(wd.norm()), 1e-12))
            assert rel < 5e-2 or float((gd - wd).abs().max()) < 1e-5, f"grad {k}: relative L2 error {rel:.3e} vs bf16 restatement"
        else:
            assert_close(got, g, rtol, atol * max(scale, 1.0), f"grad {k}")
