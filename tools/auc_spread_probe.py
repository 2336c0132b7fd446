#!/usr/bin/env python
"""How wide is the distribution of the HIP path's AUC over row orders, next to the reference's own?  One fixture, both precisions, the
default (scaled-state) and the exact replay of untouched rows, N row orders each.
    python tools/auc_spread_probe.py [fixture name] [N]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import auc_parity as AP  # noqa: E402
from oracle import cdc_oracle as O  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "auc_parity_v10k"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fx = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
cfg = fx["config"]
args = type("A", (), dict(vocab=cfg["vocab"], steps=cfg["steps"], eval_rows=cfg["eval_rows"], id_dist=cfg["id_dist"],
                          teacher_std=cfg.get("teacher_std", 0.3)))()
fd, train, ev = AP.dataset(args)
model, sd0, sha = AP.initial_state(fd)
_, yev, gev = ev
ref = np.array([v["auc"] for k, v in fx["cpu_sides"].items() if k == "ref" or k.startswith("ref_")])
c = ref.mean()
print(f"{name}: reference mean {c:.6f}, sigma {ref.std(ddof=1):.2e} over {len(ref)} runs, largest deviation {np.abs(ref - c).max():.2e}")
for precision in ("f32", "bf16"):
    for fast, scaled in ((True, True), (True, False), (False, False)):
        if os.environ.get("PROBE_ONLY") and os.environ["PROBE_ONLY"] != f"{int(fast)}{int(scaled)}":
            continue
        vals = []
        for perm in [None] + list(range(1, N)):
            p = AP.side_hip(args, fd, model, sd0, train, ev, precision, perm_seed=perm, fast_replay=fast, scaled_replay=scaled)
            vals.append(O.auc(yev, p) - c)
        v = np.array(vals)
        print(f"hip_{precision} fast_replay={fast} scaled={scaled}: mean {v.mean():+.2e} sigma {v.std(ddof=1):.2e} | " + " ".join(f"{x:+.1e}" for x in v), flush=True)
