"""AUC parity at the scale SURVEY.md 8d prescribes (fast variant): PLE-3 with the reference's dims, 26 fields x vocab 10k,
emb_dim 16, 488 training steps of 4096 rows, 0.5 M evaluation rows, dropout 0, same synthetic data and initial tensors on
both sides.

The CPU sides were run in the build container by tools/auc_parity.py — the REFERENCE itself (/root/reference/model/ple.py driven
like run.py:481-493), the reference with every batch's rows reversed (same mathematics, other summation order), and the CPU
restatement (oracle) — and their AUC / logloss are the committed fixture tests/golden/auc_parity_v10k.json.  This test trains
the HIP path (exact-fp32 and bf16 contractions) on the same data from the same initial state and holds it to

    |AUC_hip - AUC_ref|  <=  max(1e-4, the largest CPU-vs-CPU gap at THIS scale)

The north star's 1e-4 alone is not reachable by any implementation: the reference does not reproduce itself to 1e-4 under a
change of summation order at this scale (profiles/round2/auc_parity.md: 1.4e-3 here, 1.8e-4 with Zipf ids, 5.9e-4 at
vocab 1 M) — Adam turns rounding-level differences of near-zero gradients into +-lr moves and the trajectories part.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_auc_parity_at_the_protocol_scale(cuda):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import auc_parity as AP
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "auc_parity_v10k.json")))
    cfg = fx["config"]
    args = type("A", (), dict(vocab=cfg["vocab"], steps=cfg["steps"], eval_rows=cfg["eval_rows"], id_dist=cfg["id_dist"]))()
    fd, train, ev = AP.dataset(args)
    model, sd0, sha = AP.initial_state(fd)
    if sha != fx["init_sha"]:
        pytest.skip(f"this torch build ({torch.__version__}) initialises the model differently from the one the CPU sides were run "
                    f"with ({fx['torch']}): the fixture's trajectories start elsewhere")
    _, yev, gev = ev
    ref = fx["cpu_sides"]["ref"]
    floor = fx["cpu_vs_cpu_floor"]
    band = max(1e-4, floor)
    report = {}
    for precision in ("f32", "bf16"):
        p = AP.side_hip(args, fd, model, sd0, train, ev, precision)
        assert np.isfinite(p).all()
        report[precision] = {"auc": O.auc(yev, p), "logloss": O.logloss(yev, p),
                             "domain_auc": [O.auc(yev[gev == k], p[gev == k]) for k in range(3)]}
    print(f"AUC ref {ref['auc']:.6f}; cpu-vs-cpu floor at this scale {floor:.2e} (ref / ref rows reversed / oracle); "
          + "; ".join(f"hip_{k} {v['auc']:.6f} ({v['auc'] - ref['auc']:+.2e})" for k, v in report.items()))
    sides = list(fx["cpu_sides"].values())
    ll_spread = max(abs(a["logloss"] - b["logloss"]) for a in sides for b in sides)
    print(f"logloss ref {ref['logloss']:.6f}, cpu spread {ll_spread:.2e}; "
          + "; ".join(f"hip_{k} {v['logloss'] - ref['logloss']:+.2e}" for k, v in report.items()))
    for k, v in report.items():
        assert abs(v["auc"] - ref["auc"]) <= band, f"hip_{k}: |dAUC| {abs(v['auc'] - ref['auc']):.2e} > {band:.2e}"
        # logloss (run.py:690-711): three CPU sides are a small sample of the trajectory spread, so 3x their largest gap, and
        # never tighter than 5e-4 (1e-3 relative of the 0.435 logloss)
        assert abs(v["logloss"] - ref["logloss"]) <= max(5e-4, 3 * ll_spread), f"hip_{k}: logloss {v['logloss']} vs {ref['logloss']}"
        # per-domain AUCs: populations a third of the whole, so the overall band or twice the CPU sides' own spread of that domain
        for d in range(3):
            spread = max(abs(a["domain_auc"][d] - b["domain_auc"][d]) for a in sides for b in sides)
            gap = abs(v["domain_auc"][d] - ref["domain_auc"][d])
            assert gap <= max(band, 2 * spread), f"hip_{k}: domain {d} AUC gap {gap:.2e} (cpu spread {spread:.2e})"
