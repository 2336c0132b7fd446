"""Evaluation path: cdc_eval_metrics and the Run.test mirror against the reference's own numbers
(tests/golden/g9_metrics.npz: sklearn roc_auc_score / log_loss captured in the reference's environment, with ties and a
single-class domain) and against the oracle restatement on larger inputs.  AUC comes out of integer rank sums and the
loss out of a fixed-order double sum, so the tolerance is 1e-12."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _close(a, b, tol=1e-12):
    if np.isnan(b):
        return np.isnan(a)
    return abs(a - b) <= tol * max(1.0, abs(b))


def test_metrics_golden_g9(cuda):
    from cdcmdr_amd.evaluate import eval_metrics
    d = np.load(os.path.join(GOLD, "g9_metrics.npz"))
    pred = torch.from_numpy(d["scores"]).to(cuda)
    label = torch.from_numpy(d["targets"].astype(np.int16)).to(cuda)
    dom = torch.from_numpy(d["domains"].astype(np.int32)).to(cuda)
    auc, loss, rows, pos = [t.cpu().numpy() for t in eval_metrics(pred, label, dom, 4)]
    assert _close(auc[4], float(d["auc"])) and _close(loss[4], float(d["logloss"]))
    for k in range(4):
        assert _close(auc[k], float(d[f"auc_d{k}"])), (k, auc[k], float(d[f"auc_d{k}"]))
        assert _close(loss[k], float(d[f"logloss_d{k}"])), (k, loss[k])
        assert rows[k] == int((d["domains"] == k).sum()) and pos[k] == int(d["targets"][d["domains"] == k].sum())
    assert np.isnan(auc[3]) and np.isnan(loss[3])              # the single-class domain: run.py:699-704
    assert rows[4] == 500


@pytest.mark.parametrize("n,n_domain", [(1, 1), (2, 1), (70_000, 7), (300_000, 3)])
def test_metrics_against_oracle_with_ties_and_ragged_domains(cuda, n, n_domain):
    from cdcmdr_amd.evaluate import eval_metrics
    rng = np.random.default_rng(n)
    s = rng.random(n).astype(np.float32)
    s[rng.random(n) < 0.3] = np.float32(0.5)                    # one huge tie group
    s[rng.random(n) < 0.05] = np.float32(0.0)                   # clipped by the loss
    s[rng.random(n) < 0.05] = np.float32(1.0)
    if n > 10:
        s[:4] = [-0.0, 0.0, 1e-30, 1.0 - 2 ** -24]              # signed zeros tie; denormal-ish and just-below-one values
    t = (rng.random(n) < 0.3).astype(np.int16)
    dom = rng.integers(0, n_domain, size=n).astype(np.int32)
    if n_domain >= 7:
        dom[dom == 5] = 4                                       # an empty domain
        t[dom == 2] = 0                                         # a single-class domain
    X = np.zeros((n, 3), dtype=np.int32)
    X[:, 1] = dom
    Xd = torch.from_numpy(X).to(cuda)
    auc, loss, rows, pos = [v.cpu().numpy() for v in eval_metrics(torch.from_numpy(s).to(cuda), torch.from_numpy(t).to(cuda),
                                                                  Xd[:, 1] if n_domain > 1 else None, n_domain)]   # strided column
    segs = [(dom == k) for k in range(n_domain)] + [np.ones(n, dtype=bool)]
    for k, mk in enumerate(segs):
        assert rows[k] == int(mk.sum()) and pos[k] == int(t[mk].sum())
        if mk.sum() == 0 or t[mk].sum() in (0, mk.sum()):
            assert np.isnan(auc[k]) and np.isnan(loss[k])
            continue
        assert _close(auc[k], O.auc(t[mk], s[mk])), (k, auc[k], O.auc(t[mk], s[mk]))
        assert _close(loss[k], O.logloss(t[mk], s[mk]), 1e-11), (k, loss[k], O.logloss(t[mk], s[mk]))


def test_metrics_flag_bad_rows(cuda):
    from cdcmdr_amd.evaluate import eval_metrics
    s = torch.tensor([0.2, float("nan"), 0.7], device=cuda)
    t = torch.tensor([0, 1, 1], dtype=torch.int16, device=cuda)
    eval_metrics(s, t)
    assert int(eval_metrics.last_err.item()) == 2
    eval_metrics(torch.tensor([0.2, 0.3, 0.7], device=cuda), t, torch.tensor([0, 1, 5], dtype=torch.int32, device=cuda), 2)
    assert int(eval_metrics.last_err.item()) == 3


def test_evaluator_mirrors_run_test(cuda):
    """Run.test's result_dict from the HIP eval forward: predictions equal the oracle's eval forward, and every metric
    equals sklearn's definition applied to those predictions."""
    from cdcmdr_amd.evaluate import Evaluator
    from cdcmdr_amd.model.mmoe import MMoE
    FD = [30, 2000, 7, 300, 3]
    torch.manual_seed(3)
    model = MMoE(FD, 8, 3, 4, (32, 16), (8,), dropout=0.2).to(cuda).set_precision("f32")
    for m in model.modules():                                   # non-trivial running statistics
        if hasattr(m, "running_mean") and m.running_mean is not None:
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    rng = np.random.default_rng(5)
    n, bs = 2500, 1024                                          # ragged last batch
    X = np.stack([rng.integers(0, d, size=n) for d in FD], axis=1).astype(np.int32)
    y = rng.integers(0, 2, size=n).astype(np.int16)
    g = X[:, 4].astype(np.int64)
    loader = [(torch.from_numpy(X[i:i + bs]).to(cuda), torch.from_numpy(y[i:i + bs]).to(cuda).reshape(-1, 1),
               torch.from_numpy(g[i:i + bs]).to(cuda).reshape(-1, 1)) for i in range(0, n, bs)]
    w = {0: 0.5, 1: 0.3, 2: 0.2}
    ev = Evaluator(model, mode="multi", domain_idx=4, n_domain=3, domain_cnt_weight=w)
    model.train()
    res = ev.test(loader)
    assert model.training                                       # restored
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = O.mmoe_forward(sd, X, FD, 3, training=False).gather(1, torch.from_numpy(g).reshape(-1, 1)).squeeze(1).numpy()
    pred, label, dom = ev.predict(loader)
    got = pred.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    assert _close(res["total_auc"], O.auc(y, got)) and _close(res["total_loss"], O.logloss(y, got), 1e-11)
    mean_auc = mean_loss = 0
    for d in range(3):
        mk = X[:, 4] == d
        assert _close(res["domain_auc"][d], O.auc(y[mk], got[mk])) and _close(res["domain_loss"][d], O.logloss(y[mk], got[mk]), 1e-11)
        mean_auc += w[d] * O.auc(y[mk], got[mk])
        mean_loss += w[d] * O.logloss(y[mk], got[mk])
    assert abs(res["mean_auc"] - mean_auc) < 1e-12 and abs(res["mean_loss"] - mean_loss) < 1e-11
    # a label column with one class: the reference's roc_auc_score raises before any per-domain figure exists
    loader1 = [(a, torch.ones_like(b), c) for a, b, c in loader]
    with pytest.raises(ValueError):
        ev.test(loader1)
