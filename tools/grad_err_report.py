"""Per-tensor relative L2 error of the bf16 path's parameter gradients (test infrastructure: imports oracle/).

    python tools/grad_err_report.py              on a GPU box: the HIP bf16 path against the oracle's bf16 restatement with exact
                                                 and with fp32 accumulation, and against the fp32 oracle
    python tools/grad_err_report.py --cpu-floor  no GPU: the two CPU restatements against EACH OTHER (same rounded operands, only the
                                                 accumulation differs) — the distance any two correct implementations show

The bounds in tests/helpers.py::compare_param_grads come from these numbers (profiles/round2/grad_err.txt)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from helpers import O, is_pre_bn_bias, make_ids, oracle_grads  # noqa: E402
import cdcmdr_amd  # noqa: E402,F401
from cdcmdr_amd.model.mmoe import MMoE  # noqa: E402
from cdcmdr_amd.model.ple import PLE  # noqa: E402
from cdcmdr_amd.model.star import STAR  # noqa: E402

CPU_FLOOR = "--cpu-floor" in sys.argv
fd = [1000] * 26
LABEL = {"exact": "bf16 restatement, exact accumulation", True: "bf16 restatement, fp32 accumulation", False: "fp32 oracle"}


def rel_errors(got, want, names):
    errs = {}
    for k, g in want.items():
        if g is None or got.get(k) is None or is_pre_bn_bias(k, names):
            continue
        gd, wd = got[k].detach().cpu().double(), g.double()
        errs[k] = float((gd - wd).norm() / max(float(wd.norm()), 1e-12))
    return errs


def show(title, errs, d):
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    v = np.array(list(errs.values()))
    if os.environ.get("CDC_REPORT_CLASSES"):
        import re
        cls = {}
        for k, e in errs.items():
            cls.setdefault(re.sub(r"\.\d+\.", ".N.", k), []).append(e)
        print("      tower 0/1: " + ", ".join(f"{k} {e:.1e}" for k, e in errs.items() if ".0." in k[:16] or ".1." in k[:16] or k.startswith("shared")))
        print("      by class: " + ", ".join(f"{c} {np.median(v):.1e}" for c, v in sorted(cls.items())))
    print(f"{title}: gradient tensors max {v.max():.3e}  p90 {np.quantile(v, 0.9):.3e}  median {np.median(v):.3e}; probabilities max |d| "
          f"{float(d.max()):.3e} mean |d| {float(d.mean()):.3e}\n      worst: " + ", ".join(f"{k} {e:.2e}" for k, e in worst), flush=True)


def oracle_side(mode, forward, sd, x, gout):
    O.MATMUL_BF16 = mode
    try:
        return oracle_grads(lambda s: forward(s, x), sd, gout)
    finally:
        O.MATMUL_BF16 = False


def run(name, model, forward, B, seed):
    rng = np.random.default_rng(seed)
    x = make_ids(rng, B, fd)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = set(sd)
    n_out = model.n_tower
    gout = torch.randn((B, n_out), generator=torch.Generator().manual_seed(seed))
    if CPU_FLOOR:
        ra, ga = oracle_side("exact", forward, sd, x, gout)
        rb, gb = oracle_side(True, forward, sd, x, gout)
        show(f"{name}: CPU exact-accumulation vs CPU fp32-accumulation restatement", rel_errors(gb, ga, names), (ra - rb).abs())
        return
    dev = torch.device("cuda:0")
    model = model.to(dev).set_precision("bf16")
    model.train()
    out = model(torch.from_numpy(x).to(dev))
    out.backward(gout.to(dev))
    got = {k: p.grad for k, p in model.named_parameters()}
    for mode in ("exact", True, False):
        ref, grads = oracle_side(mode, forward, sd, x, gout)
        show(f"{name}: HIP bf16 vs {LABEL[mode]}", rel_errors(got, grads, names), (out.detach().cpu() - ref).abs())


torch.manual_seed(2)
run("mmoe8 B=512", MMoE(fd, 16, 3, 8, (256, 128, 64), (64, 32), dropout=0.0),
    lambda s, x: O.mmoe_forward(s, x, fd, 3, training=True), 512, 7)
torch.manual_seed(0)
run("ple3 B=512", PLE(fd, 16, 3, 2, 2, ((256, 128), (64,)), (64, 32), dropout=0.0),
    lambda s, x: O.ple_forward(s, x, fd, 3, training=True), 512, 1)
if os.environ.get("CDC_REPORT_BIG"):
    torch.manual_seed(3)
    run("star30 B=4096", STAR(fd, 16, 30, (256, 128, 64, 32), dropout=0.0),
        lambda s, x: O.star_forward(s, x, fd, 30, training=True), 4096, 8)
    torch.manual_seed(2)
    run("mmoe8 B=4096", MMoE(fd, 16, 3, 8, (256, 128, 64), (64, 32), dropout=0.0),
        lambda s, x: O.mmoe_forward(s, x, fd, 3, training=True), 4096, 7)
    sys.exit(0)
torch.manual_seed(3)
run("star30 B=256", STAR(fd, 16, 30, (256, 128, 64, 32), dropout=0.0),
    lambda s, x: O.star_forward(s, x, fd, 30, training=True), 256, 8)
