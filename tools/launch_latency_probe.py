#!/usr/bin/env python
"""How long does a step take to START on an idle GPU?  The C2 step replayed as a hipGraph vs issued launch by launch:
host time per call, and wall time of n steps between two synchronisations (n = 1, 2, 5, 20) -> the fixed cost of a timed bracket.
    python tools/launch_latency_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd.model.ple import PLE  # noqa: E402
from cdcmdr_amd.optim import FusedAdam  # noqa: E402
from cdcmdr_amd.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
V, B = 100_000, 4096
for use_graph in (True, False):
    torch.manual_seed(0)
    with torch.device(dev):
        m = PLE([V] * 26, 16, 3, 2, 2, ((256, 128), (64,)), (64, 32), 0.2)
    m.set_precision("bf16")
    opt = FusedAdam(m, table_mode="lazy")
    ts = TrainStep(m, opt, B, mode="multi", use_graph=use_graph)
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.integers(0, V, size=(64, B, 26)).astype(np.int32)).to(dev)
    X[:, :, 10] %= 3
    y = torch.from_numpy(rng.integers(0, 2, size=(64, B)).astype(np.int16)).to(dev)
    g = X[:, :, 10].long()
    k = [0]

    def run(n):
        for _ in range(n):
            j = k[0] % 64
            k[0] += 1
            ts.step(X[j], y[j], g[j], next_X=X[(j + 1) % 64])
    run(80)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(200)
    host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t0) / 200
    out = [f"graph={use_graph}: host time per step() call {host * 1e6:.0f} us, steady {steady * 1e6:.0f} us/step;"]
    for n in (1, 2, 5, 20):
        ts_ = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(n)
            torch.cuda.synchronize()
            ts_.append(time.perf_counter() - t0)
        out.append(f"{n} steps between syncs: {np.median(ts_) * 1e6:.0f} us (= {np.median(ts_) * 1e6 - n * steady * 1e6:+.0f} over steady)")
    print(" ".join(out))
