"""CDC's clustering control functions against the reference's own outputs: tests/golden/g10_*.npz were
produced by importing the reference's model/cdc.py and calling update_group three times in a row (k-means initialisation,
iterative and greedy re-assignment) for both affinity functions and both metrics, plus the helper functions on the state
each call left behind (tools/make_golden_cdc_group.py).  Host-side logic only: runs without a GPU."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def test_causal_kernel_matches_the_reference():
    from cdcmdr_amd.clustering import causal_kernel
    d = np.load(os.path.join(GOLD, "g10_causal_matrix.npz"))
    np.testing.assert_allclose(causal_kernel(d["X"]), d["kappa"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(causal_kernel(d["X"], alpha=0.05), d["kappa_alpha"], rtol=1e-12, atol=1e-13)
    k = causal_kernel(d["X"])
    assert np.allclose(np.diag(k), 1.0) and np.allclose(k, k.T) and k.max() <= 1.0


@pytest.mark.parametrize("affinity,metric", [("minus", "loss"), ("divide", "loss"), ("minus", "auc"), ("divide", "auc")])
def test_three_regroupings_match_the_reference(tmp_path, monkeypatch, affinity, metric):
    import sklearn
    from cdcmdr_amd.model.cdc import CDC
    d = np.load(os.path.join(GOLD, f"g10_cdc_group_{affinity}_{metric}.npz"))
    if str(d["sklearn"]) != sklearn.__version__:
        pytest.skip(f"k-means fixture was captured with scikit-learn {d['sklearn']}")
    monkeypatch.chdir(tmp_path)
    n_domain, n_cluster, n_mask = 9, 3, 7
    cfg = types.SimpleNamespace(mmoe_n_expert=2, ple_n_expert_specific=1, ple_n_expert_shared=1, gate_hidden_dim=8,
                                dataset_name="golden", p_weight=0.5, p_weight_method="linear_decay", p_weight_exp_decay=0.9,
                                old_matrix_weight=0.3, affinity_func=affinity, use_atten=False, n_cross_layers=3)
    torch.manual_seed(1)
    cdc = CDC([7, 100, 3, 50, n_domain, 29], 4, n_cluster, n_domain, "mmoe", (8,), (4,), 4, domain_cnt_weight=d["cnt_w"],
              n_causal_mask=n_mask, use_metric=metric, dropout=0.0, config=cfg)
    np.random.seed(123)                                           # KMeans draws from numpy's global generator
    for call, mode in enumerate(["iterative", "iterative", "greedy"]):
        cdc.matrix_A = torch.from_numpy(d[f"in{call}/A"].copy())
        cdc.matrix_B = torch.from_numpy(d[f"in{call}/B"].copy())
        cdc.matrix_mask = torch.from_numpy(d[f"in{call}/mask"].copy())
        got = cdc.update_group(mode=mode)
        assert list(got) == d[f"out{call}/domain2group_list"].tolist(), f"call {call}: assignment"
        assert [[int(v) for v in g] for g in cdc.s_group2domain_list] == json.loads(str(d[f"out{call}/s_group2domain"]))
        assert [[int(v) for v in g] for g in cdc.t_group2domain_list] == json.loads(str(d[f"out{call}/t_group2domain"]))
        np.testing.assert_allclose(cdc.matrix_causal.numpy(), d[f"out{call}/matrix_causal"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(cdc.matrix_A.numpy(), d[f"out{call}/matrix_A"])
        np.testing.assert_array_equal(cdc.matrix_B.numpy(), d[f"out{call}/matrix_B"])
        np.testing.assert_array_equal(np.asarray(cdc.matrix_mask), d[f"out{call}/matrix_mask"])
        assert cdc.p_weight == float(d[f"out{call}/p_weight"])
        grp = [0, 3, 5, 8]
        np.testing.assert_allclose(cdc.calc_domain_lambda_in_group(group=grp).numpy(), d[f"out{call}/lambda_grp_all"], rtol=1e-6)
        np.testing.assert_allclose(cdc.calc_domain_lambda_in_group(group=grp, domain=[1, 3, 4]).numpy(), d[f"out{call}/lambda_grp_sub"], rtol=1e-6)
        assert cdc.get_center_domain_in_group([1, 2, 4, 6, 7], center_num=2) == d[f"out{call}/center2"].tolist()
        assert abs(float(cdc.calc_metric_in_source_group(2, [0, 1, 5])) - float(d[f"out{call}/metric_d2"])) < 1e-6
        assert cdc.get_source_domain([1, 4, 7], group_idx=1) == d[f"out{call}/source_of_147"].tolist()
    assert cdc.call_update_group == 3
    assert os.path.exists(os.path.join("result", "golden", "matrix_A step-3.csv"))
