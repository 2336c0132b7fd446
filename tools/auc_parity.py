#!/usr/bin/env python
"""The AUC-parity protocol of SURVEY.md 8d, at its stated scale.

    PLE 3-domain (expert dims ((256,128),(64,)), towers (64,32): config.py:39-42), 26 fields x vocab V, emb_dim 16,
    N_train = 488 steps x 4096 rows, N_eval = 0.5 M rows, dropout 0 (torch's dropout stream cannot be reproduced),
    synthetic ids + planted teacher (cdcmdr_amd/synth.py, numpy PCG64: the same data on every machine),
    the same initial tensors on every side (torch.manual_seed(2000) on the CPU, then copied).

Sides (each trains from the same state on the same batches in the same order, then predicts the evaluation rows):
    ref          the REFERENCE itself: /root/reference/model/ple.py driven exactly like run.py:481-493 (BCELoss on the gathered
                 tower, + get_regularization_loss, zero_grad, backward, torch.optim.Adam(lr 1e-3, betas (0.9,0.99), eps 1e-8,
                 weight_decay 1e-8) on every parameter).  Only where /root/reference is mounted (the build container).
    ref_rev      the same with the rows of every batch reversed: identical mathematics, other summation order — the
                 reference's own reproducibility floor at this scale.
    ref_perm<k>  the same with the rows of every batch in a seeded random order (k = 1, 2, ...): further samples of that
                 floor, so that the band the HIP sides are held to is a distribution and not one pair.
    oracle       oracle/cdc_oracle.py (the CPU restatement) with the same loop.
    oracle_bf16[_perm<k>]  the restatement with the operands of every contraction rounded to bf16 where the kernels round them
                 (fp32 accumulate): the CPU statement of the bf16 path's arithmetic, plain and with reordered batches — the
                 spread the bf16 HIP side is entitled to on top of the reference's own.
    hip_f32 / hip_bf16   the HIP path (TrainStep, lazy table, hipGraph) with exact-fp32 / bf16 contractions.  Needs a GPU.

Every side writes <out>/<side>.npy (float32 predictions of the evaluation rows' own tower); `--summarise` turns whatever
is there into <out>/summary.json (AUC, logloss, per-domain AUC, deltas against `ref`, the floor).

    python tools/auc_parity.py --vocab 10000 --sides ref,ref_rev,oracle --out gpurun_out/auc_v10k        (CPU, here)
    python tools/auc_parity.py --vocab 10000 --sides hip_f32,hip_bf16 --out gpurun_out/auc_v10k          (GPU box)
    python tools/auc_parity.py --vocab 10000 --out gpurun_out/auc_v10k --summarise
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CDC_REFERENCE", "/root/reference")

F, D, B, N_DOMAIN, DOMAIN_IDX = 26, 16, 4096, 3, 10
EXPERT_DIMS, TOWER_DIMS = ((256, 128), (64,)), (64, 32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=488)
    ap.add_argument("--eval-rows", type=int, default=500_000)
    ap.add_argument("--sides", default="")
    ap.add_argument("--out", required=True)
    ap.add_argument("--summarise", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--id-dist", default="uniform")
    ap.add_argument("--teacher-std", type=float, default=0.3)
    ap.add_argument("--fixture", default="", help="with --summarise: also write the CPU sides as a test fixture (tests/golden/auc_parity_*.json)")
    return ap.parse_args()


def dataset(args):
    from cdcmdr_amd.synth import make_dataset
    fd = [args.vocab] * F
    n = B * args.steps + args.eval_rows
    X, y = make_dataset(n, fd, n_domain=N_DOMAIN, domain_idx=DOMAIN_IDX, seed=2000, dist=args.id_dist,
                        teacher_std=float(getattr(args, "teacher_std", 0.3)))
    ntr = B * args.steps
    g = X[:, DOMAIN_IDX].astype(np.int64)
    return fd, (X[:ntr], y[:ntr], g[:ntr]), (X[ntr:], y[ntr:], g[ntr:])


def initial_state(fd):
    """The build's own initialiser on the CPU; every side loads these tensors."""
    import torch
    from cdcmdr_amd.model.ple import PLE
    torch.manual_seed(2000)
    model = PLE(fd, D, N_DOMAIN, 2, 2, EXPERT_DIMS, TOWER_DIMS, dropout=0.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    return model, sd, h.hexdigest()[:16]


def side_ref(args, fd, sd0, train, ev, reverse, perm_seed=None):
    import tempfile
    import torch
    if not os.path.isdir(REF):
        raise SystemExit(f"the reference is not mounted at {REF}: the `ref` sides run in the build container only")
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="cdc_auc_"))
    from model.ple import PLE as RefPLE                      # the reference's own model/ple.py
    os.chdir(cwd)
    model = RefPLE(np.array(fd), D, N_DOMAIN, 2, 2, EXPERT_DIMS, TOWER_DIMS, dropout=0.0, config=None)
    missing = model.load_state_dict(sd0, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    crit = torch.nn.BCELoss()
    opt = torch.optim.Adam(params=model.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)   # run.py:720-721
    Xtr, ytr, gtr = train
    model.train()
    t0 = time.time()
    for s in range(args.steps):
        sl = slice(s * B, (s + 1) * B)
        xs, ys, gs = Xtr[sl], ytr[sl], gtr[sl]
        if reverse:
            xs, ys, gs = xs[::-1].copy(), ys[::-1].copy(), gs[::-1].copy()
        if perm_seed is not None:                                            # same rows, another order inside the batch
            pi = np.random.Generator(np.random.PCG64(1_000_003 * perm_seed + s)).permutation(B)
            xs, ys, gs = xs[pi].copy(), ys[pi].copy(), gs[pi].copy()
        X = torch.from_numpy(xs)
        y = torch.from_numpy(ys).reshape(-1, 1)
        group = torch.from_numpy(gs).reshape(-1, 1)
        pred = model(X)                                                      # run.py:481-493
        loss = crit(pred.gather(1, group).squeeze(1), y.squeeze(1).float())
        loss = loss + model.get_regularization_loss("cpu")
        model.zero_grad()
        loss.backward()
        opt.step()
        if s % 50 == 0:
            print(f"[ref{'_rev' if reverse else ''}] step {s} loss {float(loss.sum()):.5f} ({time.time() - t0:.0f} s)", flush=True)
    model.eval()
    out = []
    Xev, _, gev = ev
    with torch.no_grad():
        for i in range(0, len(Xev), 16384):
            p = model(torch.from_numpy(Xev[i:i + 16384]))
            out.append(p.gather(1, torch.from_numpy(gev[i:i + 16384]).reshape(-1, 1)).squeeze(1).numpy())
    return np.concatenate(out).astype(np.float32)


def side_oracle(args, fd, sd0, train, ev, bf16=False, perm_seed=None):
    """bf16: the restatement rounds the operands of every contraction to bf16 where the kernels do (fp32 accumulate) — the CPU
    statement of the arithmetic the bf16 HIP path runs; perm_seed: rows of every batch in a seeded random order."""
    import torch
    from oracle import cdc_oracle as O
    O.MATMUL_BF16 = bool(bf16)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd0.items() if v.dtype.is_floating_point and "running_" not in k}
    sd = dict(sd0)
    sd.update(leaves)
    l2 = {n: 1e-5 for n in O.reg_names(list(sd), "ple")}
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    Xtr, ytr, gtr = train
    t0 = time.time()
    for s in range(args.steps):
        sl = slice(s * B, (s + 1) * B)
        xs, ys, gs = Xtr[sl], ytr[sl], gtr[sl]
        if perm_seed is not None:
            pi = np.random.Generator(np.random.PCG64(1_000_003 * perm_seed + s)).permutation(B)
            xs, ys, gs = xs[pi].copy(), ys[pi].copy(), gs[pi].copy()
        stats = {}
        p = O.ple_forward(sd, xs, fd, N_DOMAIN, training=True, stats_out=stats)
        p = p.gather(1, torch.from_numpy(gs).reshape(-1, 1)).squeeze(1)
        loss = O.bce_mean(p, torch.from_numpy(ys.astype(np.float32))) + O.reg_loss(sd, l2).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
        sd.update(stats)
        if s % 50 == 0:
            print(f"[oracle] step {s} loss {float(loss):.5f} ({time.time() - t0:.0f} s)", flush=True)
    out = []
    Xev, _, gev = ev
    with torch.no_grad():
        sde = {k: v.detach() for k, v in sd.items()}
        for i in range(0, len(Xev), 16384):
            p = O.ple_forward(sde, Xev[i:i + 16384], fd, N_DOMAIN, training=False)
            out.append(p.gather(1, torch.from_numpy(gev[i:i + 16384]).reshape(-1, 1)).squeeze(1).numpy())
    O.MATMUL_BF16 = False
    return np.concatenate(out).astype(np.float32)


def side_hip(args, fd, model, sd0, train, ev, precision, perm_seed=None, fast_replay=True, scaled_replay=True, **step_kw):
    import torch
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    dev = torch.device("cuda", 0)
    model.load_state_dict(sd0)
    model = model.to(dev).set_precision(precision)
    model.train()
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, table_mode="lazy", fast_replay=fast_replay)
    if not scaled_replay:
        opt.replay_tab = None                      # (probe: the hardware rcp / sqrt at the exact recurrence's rounding points, csrc/common.h adam_elem_fast_pk)
    ts = TrainStep(model, opt, B, mode="multi", use_graph=True, **step_kw)
    Xtr, ytr, gtr = (torch.from_numpy(a).to(dev) for a in train)
    for s in range(args.steps):
        sl = slice(s * B, (s + 1) * B)
        if perm_seed is None:
            ts.step(Xtr[sl], ytr[sl], gtr[sl])
        else:                                                                # same rows, another order inside the batch (as ref_perm<k>)
            pi = torch.from_numpy(np.random.Generator(np.random.PCG64(1_000_003 * perm_seed + s)).permutation(B)).to(dev)
            ts.step(Xtr[sl][pi].contiguous(), ytr[sl][pi].contiguous(), gtr[sl][pi].contiguous())
    ts.check_ids()
    opt.flush_table()
    model.eval()
    Xev, _, gev = ev
    out = []
    with torch.no_grad():
        for i in range(0, len(Xev), 16384):
            p = model(torch.from_numpy(Xev[i:i + 16384]).to(dev))
            out.append(p.gather(1, torch.from_numpy(gev[i:i + 16384]).to(dev).reshape(-1, 1)).squeeze(1).float().cpu().numpy())
    model.to("cpu")
    return np.concatenate(out).astype(np.float32)


def summarise(args, ev, init_hash=None):
    from oracle import cdc_oracle as O
    _, yev, gev = ev
    res = {"config": {"model": "PLE-3 ((256,128),(64,)) towers (64,32)", "fields": F, "vocab": args.vocab, "emb_dim": D, "batch": B,
                      "steps": args.steps, "eval_rows": int(len(yev)), "dropout": 0.0, "id_dist": args.id_dist},
           "sides": {}}
    preds = {}
    names = ["ref", "ref_rev"] + sorted(f[:-4] for f in os.listdir(args.out) if f.startswith("ref_perm") and f.endswith(".npy"))
    names += sorted(f[:-4] for f in os.listdir(args.out) if f.startswith("oracle_bf16") and f.endswith(".npy"))
    hips = sorted(f[:-4] for f in os.listdir(args.out) if f.startswith("hip_") and f.endswith(".npy"))
    for name in names + ["oracle"] + hips:
        path = os.path.join(args.out, name + ".npy")
        if os.path.exists(path):
            preds[name] = np.load(path)
    for name, p in preds.items():
        d = {"auc": O.auc(yev, p), "logloss": O.logloss(yev, p)}
        d["domain_auc"] = [O.auc(yev[gev == k], p[gev == k]) for k in range(N_DOMAIN)]
        meta = os.path.join(args.out, name + ".json")
        if os.path.exists(meta):
            d.update(json.load(open(meta)))
        res["sides"][name] = d
    base = "ref" if "ref" in preds else ("oracle" if "oracle" in preds else None)
    if base:
        res["baseline_side"] = base
        for name in preds:
            if name != base:
                res["sides"][name]["auc_minus_" + base] = res["sides"][name]["auc"] - res["sides"][base]["auc"]
                res["sides"][name]["max_abs_pred_diff_vs_" + base] = float(np.abs(preds[name] - preds[base]).max())
        if "ref_rev" in preds and base == "ref":
            res["cpu_vs_cpu_floor"] = abs(res["sides"]["ref_rev"]["auc"] - res["sides"]["ref"]["auc"])
        reorder = [n for n in preds if n == "ref" or n.startswith("ref_")]
        if len(reorder) >= 3:
            # the reference against itself under row reorderings of every batch: the distribution the band comes from
            aucs = np.array([res["sides"][n]["auc"] for n in reorder])
            lls = np.array([res["sides"][n]["logloss"] for n in reorder])
            dom = np.array([res["sides"][n]["domain_auc"] for n in reorder])
            res["ref_reorderings"] = {"sides": reorder, "auc_mean": float(aucs.mean()), "auc_max_dev": float(np.abs(aucs - aucs.mean()).max()),
                                      "auc_std": float(aucs.std(ddof=1)), "logloss_mean": float(lls.mean()),
                                      "logloss_max_dev": float(np.abs(lls - lls.mean()).max()),
                                      "domain_auc_mean": dom.mean(0).tolist(),
                                      "domain_auc_max_dev": np.abs(dom - dom.mean(0)).max(0).tolist()}
    path = os.path.join(args.out, "summary.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1))
    if getattr(args, "fixture", ""):
        cpu = {k: {f: v[f] for f in ("auc", "logloss", "domain_auc")} for k, v in res["sides"].items() if not k.startswith("hip_")}
        any_side = next(iter(res["sides"].values()))
        fx = {"config": dict(res["config"], teacher_std=float(getattr(args, "teacher_std", 0.3))),
              "init_sha": any_side.get("init_sha"), "torch": any_side.get("torch"), "cpu_sides": cpu,
              "cpu_vs_cpu_floor": res.get("cpu_vs_cpu_floor"), "ref_reorderings": res.get("ref_reorderings"),
              "bf16_restatement_runs": sorted(k for k in cpu if k.startswith("oracle_bf16")),
              "made_by": "tools/auc_parity.py --sides ref,ref_rev,ref_perm1..N,oracle[,oracle_bf16..] (build container, imports /root/reference/model/ple.py) "
                         "+ --summarise --fixture"}
        json.dump(fx, open(args.fixture, "w"), indent=1)
        print("wrote", args.fixture)
    return res


def main():
    args = parse()
    os.makedirs(args.out, exist_ok=True)
    import torch
    if args.threads <= 0:
        try:
            args.threads = len(os.sched_getaffinity(0))
        except AttributeError:
            args.threads = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(args.threads, 64)))
    fd, train, ev = dataset(args)
    if args.summarise:
        summarise(args, ev)
        return
    model, sd0, init_hash = initial_state(fd)
    for side in [s for s in args.sides.split(",") if s]:
        t0 = time.time()
        if side == "ref":
            p = side_ref(args, fd, sd0, train, ev, reverse=False)
        elif side == "ref_rev":
            p = side_ref(args, fd, sd0, train, ev, reverse=True)
        elif side.startswith("ref_perm"):
            p = side_ref(args, fd, sd0, train, ev, reverse=False, perm_seed=int(side[len("ref_perm"):]))
        elif side == "oracle":
            p = side_oracle(args, fd, sd0, train, ev)
        elif side.startswith("oracle_bf16"):
            tail = side[len("oracle_bf16"):]
            p = side_oracle(args, fd, sd0, train, ev, bf16=True, perm_seed=int(tail[5:]) if tail.startswith("_perm") else None)
        elif side.startswith("hip_f32") or side.startswith("hip_bf16"):
            prec, _, tail = side[4:].partition("_perm")
            p = side_hip(args, fd, model, sd0, train, ev, prec, perm_seed=int(tail) if tail else None)
        else:
            raise SystemExit(f"unknown side {side}")
        np.save(os.path.join(args.out, side + ".npy"), p)
        json.dump({"init_sha": init_hash, "seconds": round(time.time() - t0, 1), "threads": torch.get_num_threads(),
                   "torch": torch.__version__}, open(os.path.join(args.out, side + ".json"), "w"))
        print(f"[{side}] done in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
