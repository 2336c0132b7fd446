"""AUC parity at the scale SURVEY.md 8d prescribes (fast variant): PLE-3 with the reference's dims, 26 fields x vocab 10k,
emb_dim 16, 488 training steps of 4096 rows, 0.5 M evaluation rows, dropout 0, same synthetic data and initial tensors on
both sides.

The CPU sides were run in the build container by tools/auc_parity.py and are committed as fixtures
(tests/golden/auc_parity_*.json): the REFERENCE itself (/root/reference/model/ple.py driven like run.py:481-493), the
reference with the rows of every batch reversed and in five seeded random orders (same mathematics, other summation
order: SEVEN runs of the reference against itself), and the CPU restatement (oracle).  Three data sets:

    v10k          uniform ids, teacher std 0.3   reference AUC 0.521  (near chance: says little about the model, a lot about
                                                                       how far two correct runs drift apart)
    v10k_zipf     Zipf ids,    teacher std 0.3   reference AUC 0.672
    v10k_zipf_t05 Zipf ids,    teacher std 0.5   reference AUC 0.733

This test trains the HIP path (exact-fp32 and bf16 contractions) on the same data from the same initial state and holds
BOTH precisions to the band the reference establishes for itself:

    |AUC_hip - mean(AUC of the reference's reorderings)|  <=  max(1e-4, max deviation of a reordering from that mean)

The north star's 1e-4 alone is not reachable by any implementation: the reference does not reproduce itself to 1e-4 under
a change of summation order at this scale (max deviation from the mean: 6.9e-4 / 4.6e-4 on the Zipf sets) — Adam turns
rounding-level differences of near-zero gradients into +-lr moves and the trajectories part.  The band is NOT widened
beyond what the seven reference runs span; the bf16 side is additionally entitled to what CPU runs of the bf16 restatement
(the same arithmetic, stated with torch CPU ops) span around the same centre, where a fixture holds such runs.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = ["auc_parity_v10k_zipf", "auc_parity_v10k_zipf_t05", "auc_parity_v10k"]
LEARNABLE = {"auc_parity_v10k_zipf": 0.65, "auc_parity_v10k_zipf_t05": 0.70}       # the teacher must be learnable on these sets


@pytest.mark.parametrize("name", FIXTURES)
def test_auc_parity_at_the_protocol_scale(cuda, name):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import auc_parity as AP
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    cfg = fx["config"]
    args = type("A", (), dict(vocab=cfg["vocab"], steps=cfg["steps"], eval_rows=cfg["eval_rows"], id_dist=cfg["id_dist"],
                              teacher_std=cfg.get("teacher_std", 0.3)))()
    fd, train, ev = AP.dataset(args)
    model, sd0, sha = AP.initial_state(fd)
    # the fixture's trajectories start from these tensors: another initial state is a different experiment, not a skip
    assert sha == fx["init_sha"], (f"this torch build ({torch.__version__}) initialises the model differently from the one the CPU sides "
                                   f"were run with ({fx['torch']}): regenerate {name}.json with tools/auc_parity.py")
    _, yev, gev = ev
    ref = fx["cpu_sides"]["ref"]
    if name in LEARNABLE:
        assert ref["auc"] > LEARNABLE[name], "AUC parity proves nothing if the teacher is not learnable"
    ro = fx.get("ref_reorderings")
    sides = [v for k, v in fx["cpu_sides"].items() if k == "ref" or k.startswith("ref_")]
    if ro:
        centre, band = ro["auc_mean"], max(1e-4, ro["auc_max_dev"])
        ll_centre, ll_band = ro["logloss_mean"], ro["logloss_max_dev"]
        dom_centre, dom_band = ro["domain_auc_mean"], ro["domain_auc_max_dev"]
    else:                                   # (a fixture without the reordered runs: the largest CPU-vs-CPU gap)
        centre, band = ref["auc"], max(1e-4, fx["cpu_vs_cpu_floor"])
        ll_centre, ll_band = ref["logloss"], max(abs(a["logloss"] - b["logloss"]) for a in sides for b in sides)
        dom_centre = ref["domain_auc"]
        dom_band = [max(abs(a["domain_auc"][d] - b["domain_auc"][d]) for a in sides for b in sides) for d in range(3)]
    # the bf16 path's own entitlement: CPU runs of the bf16 RESTATEMENT (operands of every contraction rounded to bf16 where the
    # kernels round them; plain and with reordered batches) where the fixture holds them — bf16 rounding is a larger
    # perturbation of the trajectory than a change of summation order, and the HIP bf16 side is held to what the reference's
    # reorderings AND those CPU bf16 runs span around the same centre
    bf = [v for k, v in fx["cpu_sides"].items() if k.startswith("oracle_bf16")]
    band_bf = max([band] + [abs(v["auc"] - centre) for v in bf])
    ll_band_bf = max([ll_band] + [abs(v["logloss"] - ll_centre) for v in bf])
    dom_band_bf = [max([dom_band[d_]] + [abs(v["domain_auc"][d_] - dom_centre[d_]) for v in bf]) for d_ in range(3)]
    report, runs = {}, {}
    # on the Zipf set every precision is trained THREE times (the batches as they are, and in two seeded row orders): one run is one
    # sample of the trajectory distribution; the mean over the orders is held to half the band (profiles/round3/auc_parity_v10k_zipf.json:
    # over five orders the HIP means sit 1e-5 (fp32) and 7e-5 (bf16) from the reference's mean)
    orders = [None, 1, 2] if name == "auc_parity_v10k_zipf" else [None]
    for precision in ("f32", "bf16"):
        runs[precision] = []
        for perm in orders:
            p = AP.side_hip(args, fd, model, sd0, train, ev, precision, perm_seed=perm)
            assert np.isfinite(p).all()
            runs[precision].append({"auc": O.auc(yev, p), "logloss": O.logloss(yev, p),
                                    "domain_auc": [O.auc(yev[gev == k], p[gev == k]) for k in range(3)]})
        report[precision] = runs[precision][0]
    print(f"{name}: AUC ref {ref['auc']:.6f}, mean of {len(sides)} reference reorderings {centre:.6f} +- {band:.2e}; "
          + "; ".join(f"hip_{k} {v['auc']:.6f} ({v['auc'] - centre:+.2e} from the mean, {v['auc'] - ref['auc']:+.2e} from ref)"
                      for k, v in report.items()))
    print(f"{name}: logloss mean {ll_centre:.6f} +- {ll_band:.2e}; " + "; ".join(f"hip_{k} {v['logloss'] - ll_centre:+.2e}" for k, v in report.items()))
    for k in report:
        band, ll_band, dom_band = (band_bf, ll_band_bf, dom_band_bf) if k == "bf16" else (band, ll_band, dom_band)
        for i, v in enumerate(runs[k]):
            tag = f"hip_{k} (row order {orders[i]})"
            assert abs(v["auc"] - centre) <= band, f"{tag}: |AUC - mean of the reference's reorderings| {abs(v['auc'] - centre):.2e} > {band:.2e}"
            # logloss (run.py:690-711): twice the reorderings' own largest deviation (seven runs are a small sample of the spread),
            # never tighter than 2e-4 (5e-4 relative of a 0.39-0.44 logloss)
            assert abs(v["logloss"] - ll_centre) <= max(2e-4, 2 * ll_band), f"{tag}: logloss {v['logloss']} vs {ll_centre}"
            # per-domain AUCs: populations a third of the whole and a single run: 2.5 x the largest deviation the CPU runs show for
            # that domain (measured over ten HIP runs on the Zipf set: up to 2.1 x), or the overall band
            for d in range(3):
                gap = abs(v["domain_auc"][d] - dom_centre[d])
                assert gap <= max(band, 2.5 * dom_band[d]), f"{tag}: domain {d} AUC gap {gap:.2e} (CPU runs deviate {dom_band[d]:.2e})"
        if len(runs[k]) > 1:
            mean_auc = float(np.mean([v["auc"] for v in runs[k]]))
            print(f"{name}: hip_{k} mean over {len(runs[k])} row orders {mean_auc:.6f} ({mean_auc - centre:+.2e} from the reference's mean)")
            assert abs(mean_auc - centre) <= band / 2, f"hip_{k}: the mean over {len(runs[k])} row orders is {mean_auc - centre:+.2e} from the reference's mean"
            for d in range(3):
                md = float(np.mean([v["domain_auc"][d] for v in runs[k]]))
                assert abs(md - dom_centre[d]) <= max(band / 2, 1.5 * dom_band[d]), f"hip_{k}: mean domain {d} AUC {md - dom_centre[d]:+.2e}"
