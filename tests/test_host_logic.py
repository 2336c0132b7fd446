"""CPU-only checks of host-side logic: the Adam scalar table, the rounding pattern of the per-element Adam routine
(its plain-C restatement oracle/adam_elem_ref.c vs torch.optim.Adam on the CPU, bit for bit), the synthetic data
generator, the gradient-interval bookkeeping of the plan builder."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_adam_ref():
    out_dir = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libadam_elem_ref.so")
    src = os.path.join(ROOT, "oracle", "adam_elem_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src, "-lm"], check=True)
    lib = C.CDLL(so)
    lib.adam_elem_ref.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 8
    return lib


def test_step_scalar_table_is_torchs_double_arithmetic_rounded_once():
    from cdcmdr_amd.optim import step_scalar_table
    tab = step_scalar_table(1e-3, 0.9, 0.99, n=3000)
    for t in (1, 2, 10, 157, 1000, 2999):
        assert tab[t, 0].item() == np.float32(1e-3 / (1 - 0.9 ** t))
        assert tab[t, 1].item() == np.float32(math.sqrt(1 - 0.99 ** t))
    # both have converged to their limits long before the table ends: clamping the index is exact
    assert tab[-1, 0].item() == np.float32(1e-3) and tab[-1, 1].item() == np.float32(1.0)
    assert torch.equal(tab[2500], tab[2999])


def test_adam_element_routine_reproduces_torch_cpu_adam_bits():
    """The C restatement shares its expression sequence with adam_elem() in csrc/common.h.  Touched elements (batch
    gradient + L2) and untouched ones (L2 only) over five steps: >= 99.9 % of the weights bit-identical,
    the rest within 2 ulp (ATen's vector/scalar-tail split is not reproducible element for element)."""
    lib = _build_adam_ref()
    torch.manual_seed(0)
    n = 100_000
    w0 = torch.randn(n)
    g_batch = torch.randn(n) * 0.01
    g_batch[::2] = 0.0
    lr, b1, b2, eps, wd, l2 = 1e-3, 0.9, 0.99, 1e-8, 1e-8, 1e-5
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    w, m, v = w0.numpy().copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    f32 = lambda x: float(np.float32(x))  # noqa: E731
    for t in range(1, 6):
        loss = (p * g_batch).sum() + torch.sum(l2 * torch.square(p))
        opt.zero_grad()
        loss.backward()
        opt.step()
        g = g_batch.numpy().copy()
        lib.adam_elem_ref(w.ctypes.data, m.ctypes.data, v.ctypes.data, g.ctypes.data, n, f32(1 - b1), f32(b2), f32(1 - b2), f32(eps),
                          f32(wd), 2 * f32(l2), f32(lr / (1 - b1 ** t)), f32(math.sqrt(1 - b2 ** t)))
    ref = p.detach().numpy()
    same = (w == ref).mean()
    assert same >= 0.999, f"only {same:.5f} of the weights are bit-identical to torch's"
    ulp = np.abs(w.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
    assert ulp <= 2
    st = opt.state[p]
    assert (m == st["exp_avg"].numpy()).mean() >= 0.999 and (v == st["exp_avg_sq"].numpy()).mean() >= 0.999
    # untouched elements moved by ~lr per step towards zero
    moved = np.abs(w[::2] - w0.numpy()[::2])
    big = np.abs(w0.numpy()[::2]) > 0.1
    assert moved[big].min() > 4.5e-3 and moved[big].max() < 5.5e-3


def test_device_routine_text_matches_the_c_restatement():
    """Guards the link between the pinned C restatement and the device routine: same operations in the same order."""
    dev = open(os.path.join(ROOT, "causal-domain-clustering-for-multi-domain-recommendation_amd", "csrc", "common.h")).read()
    body = dev[dev.index("__device__ __forceinline__ void adam_elem"):dev.index("__device__ __forceinline__ AdamConsts make_consts")]
    for frag in ["__fadd_rn(g_in, __fmul_rn(c.l2_twice, w))", "fmaf(w, c.wd, g)", "fmaf(c.lerp_w, __fsub_rn(g, m), m)",
                 "__fmul_rn(v, c.beta2)", "fmaf(__fmul_rn(c.omb2, g), g, v)", "__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), c.eps",
                 "__fdiv_rn(__fmul_rn(-step_size, m), denom)"]:
        assert frag in body, frag


def test_synthetic_data_is_deterministic_and_well_formed():
    from cdcmdr_amd.synth import make_dataset
    fd = [1000] * 12
    fd[10] = 3
    X1, y1 = make_dataset(5000, fd, n_domain=3, domain_idx=10, seed=2000)
    X2, y2 = make_dataset(5000, fd, n_domain=3, domain_idx=10, seed=2000)
    assert X1.dtype == np.int32 and y1.dtype == np.int16                     # run.py:198-199
    assert np.array_equal(X1, X2) and np.array_equal(y1, y2)
    assert X1.min() >= 0 and all(X1[:, f].max() < fd[f] for f in range(12))
    assert 0.02 < y1.mean() < 0.5                                            # the planted teacher is informative, not constant
    Xz, _ = make_dataset(5000, fd, n_domain=3, domain_idx=10, seed=2000, dist="zipf")
    assert np.bincount(Xz[:, 0]).max() > 5 * np.bincount(X1[:, 0]).max()     # hot rows


def test_plan_gradient_interval_bookkeeping():
    from cdcmdr_amd.plan import Buf, _GradState
    root = torch.zeros(4, 32)
    gs = _GradState()
    whole = Buf(root, 4, 32)
    a, b, c = whole.slice(0, 8), whole.slice(8, 16), whole.slice(16, 32)
    assert gs.claim(a) is False and gs.claim(a) is True                      # first writer stores, the next accumulates
    assert not gs.is_set(whole)
    assert gs.claim(b) is False and gs.claim(c) is False
    assert gs.is_set(whole) and gs.claim(whole) is True                      # adjacent slices merge into the whole buffer
    gs2 = _GradState()
    gs2.claim(whole.slice(0, 8))
    with pytest.raises(RuntimeError):
        gs2.claim(whole.slice(4, 12))                                        # a partial overlap would mix store and add
    # activation mask geometry of slices: columns [0,20) masked
    whole.mask = (1.25, 20)
    assert whole.slice(0, 8).mask == (1.25, 8) and whole.slice(16, 32).mask == (1.25, 4) and whole.slice(24, 32).mask is None
