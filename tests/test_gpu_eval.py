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


def test_evaluator_cdc_split_mode_mirrors_run_test(cuda):
    """The CDC branch of Run.test (run.py:653-661): per-domain loaders walked in the shuffled domain sequence, every batch scored by
    the tower of ITS domain's group (`model(X, mode='split', domain_i=d)`).  Predictions equal the oracle's eval forward +
    cdc_forward's column pick; the metrics equal their definitions on those predictions."""
    import types
    from cdcmdr_amd.data import make_domain_loaders
    from cdcmdr_amd.evaluate import Evaluator
    from cdcmdr_amd.model.cdc import CDC
    n_domain, n_cluster, domain_idx, bs = 6, 2, 4, 128
    fd = [7, 300, 3, 50, n_domain, 29]
    rng = np.random.default_rng(0)
    n = 1500
    Xn = np.stack([rng.integers(0, d, size=n) for d in fd], axis=1).astype(np.int32)
    yn = rng.integers(0, 2, size=(n, 1)).astype(np.int16)
    np.random.seed(1)
    torch.manual_seed(1)
    loaders, seq, w = make_domain_loaders(torch.from_numpy(Xn), torch.from_numpy(yn), bs, cuda, domain_idx, n_domain, shuffle=False)
    cfg = types.SimpleNamespace(mmoe_n_expert=3, dataset_name="t", p_weight=0.5, p_weight_method="linear_decay", p_weight_exp_decay=0.9,
                                old_matrix_weight=0.3, affinity_func="minus", use_atten=False)
    cdc = CDC(fd, 4, n_cluster, n_domain, "mmoe", (16, 8), (8,), domain_idx, domain_cnt_weight=w, n_causal_mask=3, use_metric="loss",
              device=cuda, dropout=0.2, config=cfg).to(cuda).set_precision("f32")
    d2g = [0, 1, 1, 0, 1, 0]
    cdc.domain2group.copy_(torch.tensor(d2g))
    cdc.domain2group_list = list(d2g)
    ev = Evaluator(cdc, mode="cdc", domain_idx=domain_idx, n_domain=n_domain, domain_cnt_weight=w)
    res = ev.test((loaders, seq))
    pred, label, dom = ev.predict((loaders, seq))
    assert pred.numel() == n                                   # every row of every domain exactly once
    sd = {k[len("base_model_instance."):]: v.detach().cpu() for k, v in cdc.state_dict().items() if k.startswith("base_model_instance.")}
    # the same walk on the host: domain d's rows in their original order, batch by batch
    pos = {d: 0 for d in range(n_domain)}
    rows = {d: np.nonzero(Xn[:, domain_idx] == d)[0] for d in range(n_domain)}
    order = []
    for d in seq:
        order.append(rows[d][pos[d]:pos[d] + bs])
        pos[d] += bs
    order = np.concatenate(order)
    assert np.array_equal(np.sort(order), np.arange(n))
    Xo, yo = Xn[order], yn[order, 0]
    assert np.array_equal(dom.cpu().numpy(), Xo[:, domain_idx]) and np.array_equal(label.cpu().numpy(), yo)
    base = O.mmoe_forward(sd, Xo, fd, n_cluster, training=False)
    want = base.gather(1, torch.tensor(d2g)[torch.from_numpy(Xo[:, domain_idx].astype(np.int64))].reshape(-1, 1)).squeeze(1).numpy()
    got = pred.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    assert _close(res["total_auc"], O.auc(yo, got)) and _close(res["total_loss"], O.logloss(yo, got), 1e-11)
    mean_auc = 0
    for d in range(n_domain):
        mk = Xo[:, domain_idx] == d
        assert _close(res["domain_auc"][d], O.auc(yo[mk], got[mk]))
        mean_auc += w[d] * O.auc(yo[mk], got[mk])
    assert abs(res["mean_auc"] - mean_auc) < 1e-12
