#!/usr/bin/env python
"""Captures golden vectors by IMPORTING the reference's model/ package (read-only mount at /root/reference)
on CPU and writes them as small .npz fixtures under tests/golden/.

Runs only where /root/reference exists (the build container); the fixtures — data only: inputs, parameters
and expected outputs — are committed, the reference sources never travel.  Usage:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden.py

Vector families: G1 gather, G2 per-model forward/backward, G3 three Adam steps,
G4 STAR grouped mode (1-row and empty groups), G5 CDC modes, G6 batch of one, G7 BCE clamp,
G8 dropout statistics, G9 AUC/logloss from sklearn.
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get("CDC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 2000   # the reference's default seed (main.py:20)


def _np(t):
    return t.detach().cpu().numpy().copy()      # copy: state_dict tensors alias live (in-place updated) storage


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays)")


def _ids(rng, B, field_dims):
    return np.stack([rng.integers(0, d, size=B) for d in field_dims], axis=1).astype(np.int32)


def _pack_sd(prefix, sd):
    return {f"{prefix}/{k}": _np(v) for k, v in sd.items()}


def main():
    if not os.path.isdir(REF):
        sys.exit(f"reference not mounted at {REF}; golden vectors can only be regenerated in the build container")
    sys.path.insert(0, REF)
    os.chdir(tempfile.mkdtemp(prefix="cdc_golden_"))       # CDC.__init__ creates result/<dataset>/ under cwd
    import torch
    torch.set_num_threads(1)
    from model.layer import FeaturesEmbedding
    from model.ple import PLE
    from model.mmoe import MMoE
    from model.dcn import DCN
    from model.dcnv2 import DCNv2
    from model.star import STAR
    from model.cdc import CDC

    crit = torch.nn.BCELoss()
    FD = [7, 100, 3, 50, 11, 29]          # uneven cardinalities
    D = 4

    # ---------------------------------------------------------------- G1 gather
    torch.manual_seed(SEED)
    rng = np.random.default_rng(SEED)
    fd1 = [3, 1000, 17, 1, 256, 999, 5]
    emb = FeaturesEmbedding(fd1, 8)
    x = _ids(rng, 64, fd1)
    xt = torch.from_numpy(x)
    idx = xt + xt.new_tensor(emb.offsets).unsqueeze(0)
    out = emb(xt, squeeze_dim=True)
    _save("g1_gather", field_dims=np.array(fd1), x=x, table=_np(emb.embedding_dict.weight), idx=_np(idx), out=_np(out),
          out3d=_np(emb(xt)))

    # ---------------------------------------------------------------- G2 per model
    def capture_model(name, model, x, group, y, fwd, extra=None):
        """train-mode forward + loss + grads (run.py:481-492 order), stats after, then eval-mode forward."""
        arrays = {"x": x, "y": y}
        if group is not None:
            arrays["group"] = group
        arrays.update(_pack_sd("sd", model.state_dict()))
        model.train()
        pred = fwd(model)
        bce = crit(pred, torch.from_numpy(y).reshape(-1).float())
        reg = model.get_regularization_loss(device="cpu")
        loss = bce + reg
        model.zero_grad()
        loss.backward()
        arrays["train_pred"] = _np(pred)
        arrays["bce"] = _np(bce)
        arrays["reg"] = _np(reg)
        for k, p in model.named_parameters():
            if p.grad is not None:
                arrays[f"grad/{k}"] = _np(p.grad)
        arrays.update(_pack_sd("sd_after", {k: v for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}))
        model.eval()
        with torch.no_grad():
            arrays["eval_pred"] = _np(fwd(model))
        if extra:
            arrays.update(extra)
        _save(name, **arrays)

    B = 64
    rng = np.random.default_rng(SEED + 1)
    x = _ids(rng, B, FD)
    y = rng.integers(0, 2, size=(B, 1)).astype(np.int16)
    group3 = rng.integers(0, 3, size=(B, 1)).astype(np.int64)

    def multi(model, g):
        return lambda m: m(torch.from_numpy(x)).gather(1, torch.from_numpy(g)).squeeze(1)

    torch.manual_seed(SEED)
    capture_model("g2_ple3", PLE(FD, D, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0), x, group3, y, multi(None, group3))
    torch.manual_seed(SEED)
    capture_model("g2_mmoe4", MMoE(FD, D, 3, 4, (32, 16, 8), (8, 4), dropout=0.0), x, group3, y, multi(None, group3))
    torch.manual_seed(SEED)
    capture_model("g2_mmoe8", MMoE(FD, D, 3, 8, (32, 16, 8), (8, 4), dropout=0.0), x, group3, y, multi(None, group3))
    fd13 = [11, 50, 7, 100, 3, 29, 64, 5, 17, 200, 9, 31, 13]
    x13 = _ids(rng, B, fd13)

    def single(xx):
        return lambda m: m(torch.from_numpy(xx))

    torch.manual_seed(SEED)
    capture_model("g2_dcn13", DCN(fd13, D, 3, (32, 16, 8), dropout=0.0), x13, None, y, single(x13))
    torch.manual_seed(SEED)
    capture_model("g2_dcnv2_mix", DCNv2(fd13, D, 3, (32, 16, 8), dropout=0.0, low_rank=8, num_experts=4), x13, None, y, single(x13))
    torch.manual_seed(SEED)
    capture_model("g2_dcnv2_stacked", DCNv2(fd13, D, 2, (32, 16), dropout=0.0, model_structure="stacked", low_rank=8), x13, None, y, single(x13))
    # constructor paths of DCNv2 that raise in the reference (recorded so the mirror can raise the same way)
    errs = {}
    for tag, kw in [("v2", dict(use_low_rank_mixture=False)), ("crossnet_only", dict(model_structure="crossnet_only")),
                    ("bad_structure", dict(model_structure="nope"))]:
        try:
            torch.manual_seed(SEED)
            DCNv2(fd13, D, 2, (32, 16), dropout=0.0, **kw)
            errs[tag] = np.array("")
        except Exception as e:  # noqa: BLE001
            errs[tag] = np.array(type(e).__name__)
    _save("g2_dcnv2_ctor_errors", **errs)
    # CrossNetV2 is unreachable through DCNv2 (see above) but is a public layer: pin it standalone
    from model.layer import CrossNetV2
    torch.manual_seed(SEED)
    cn2 = CrossNetV2(24, 3)
    for prm in cn2.b:
        torch.nn.init.normal_(prm, std=0.1)
    xin = torch.randn(B, 24, requires_grad=True)
    out = cn2(xin)
    gout = torch.randn(B, 24)
    out.backward(gout)
    arr = {"x": _np(xin), "out": _np(out), "gout": _np(gout), "dx": _np(xin.grad)}
    arr.update(_pack_sd("sd", cn2.state_dict()))
    for k, prm in cn2.named_parameters():
        arr[f"grad/{k}"] = _np(prm.grad)
    _save("g2_crossnetv2_layer", **arr)
    # STAR, all-towers mode (x_group=None): [B, n_tower] then gather (run.py:669 eval path and CDC)
    n_star = 5
    group5 = rng.integers(0, n_star, size=(B, 1)).astype(np.int64)
    torch.manual_seed(SEED)
    capture_model("g2_star5_all", STAR(FD, D, n_star, (32, 16, 8), dropout=0.0), x, group5, y,
                  lambda m: m(torch.from_numpy(x)).gather(1, torch.from_numpy(group5)).squeeze(1))
    # STAR 30 towers, tiny dims (config 5 shape in miniature)
    group30 = rng.integers(0, 30, size=(B, 1)).astype(np.int64)
    torch.manual_seed(SEED)
    capture_model("g2_star30_all", STAR(FD, D, 30, (16, 8), dropout=0.0), x, group30, y,
                  lambda m: m(torch.from_numpy(x)).gather(1, torch.from_numpy(group30)).squeeze(1))

    # ---------------------------------------------------------------- G4 STAR grouped mode (run.py:477-480)
    group_g = group5.copy()
    group_g[group_g == 3] = 2            # group 3 empty
    ones = np.where(group_g[:, 0] == 4)[0]
    group_g[ones[1:], 0] = 1             # group 4 has exactly one row
    torch.manual_seed(SEED)
    star = STAR(FD, D, n_star, (32, 16, 8), dropout=0.0)
    arrays = {"x": x, "y": y, "group": group_g}
    arrays.update(_pack_sd("sd", star.state_dict()))
    star.train()
    pred, yy = star(torch.from_numpy(x), torch.from_numpy(group_g), targets=torch.from_numpy(y))
    bce = crit(pred.squeeze(), yy.squeeze().float())
    reg = star.get_regularization_loss(device="cpu")
    star.zero_grad()
    (bce + reg).backward()
    arrays.update({"train_pred": _np(pred), "train_targets": _np(yy), "bce": _np(bce), "reg": _np(reg)})
    for k, p in star.named_parameters():
        if p.grad is not None:
            arrays[f"grad/{k}"] = _np(p.grad)
    arrays.update(_pack_sd("sd_after", {k: v for k, v in star.state_dict().items() if "running_" in k or "num_batches" in k}))
    star.eval()
    with torch.no_grad():
        pe, ye = star(torch.from_numpy(x), torch.from_numpy(group_g), targets=torch.from_numpy(y))
    arrays.update({"eval_pred": _np(pe), "eval_targets": _np(ye)})
    _save("g4_star5_grouped", **arrays)

    # ---------------------------------------------------------------- G3 three consecutive R1 steps with Adam
    def three_steps(name, model, fwd_pred, xs, ys, groups):
        opt = torch.optim.Adam(params=model.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
        arrays = _pack_sd("sd0", model.state_dict())
        model.train()
        for s in range(3):
            pred = fwd_pred(model, xs[s], groups[s])
            loss = crit(pred, torch.from_numpy(ys[s]).reshape(-1).float())
            bce = loss.detach().clone()
            reg = model.get_regularization_loss(device="cpu")
            loss = loss + reg
            model.zero_grad()
            loss.backward()
            opt.step()
            arrays[f"x{s}"] = xs[s]
            arrays[f"y{s}"] = ys[s]
            if groups[s] is not None:
                arrays[f"group{s}"] = groups[s]
            arrays[f"loss{s}"] = _np(loss)
            arrays[f"bce{s}"] = _np(bce)
            arrays[f"reg{s}"] = _np(reg)
            arrays.update(_pack_sd(f"sd{s + 1}", model.state_dict()))
            st = opt.state_dict()["state"]
            names = [k for k, _ in model.named_parameters()]
            for i, k in enumerate(names):
                if i in st:
                    arrays[f"m{s + 1}/{k}"] = _np(st[i]["exp_avg"])
                    arrays[f"v{s + 1}/{k}"] = _np(st[i]["exp_avg_sq"])
        _save(name, **arrays)

    fd_sparse = [7, 400, 3, 50, 11, 29]   # field 1 has 400 ids: most rows are never touched in 3 x 32 samples
    xs = [_ids(rng, 32, fd_sparse) for _ in range(3)]
    for xx in xs:
        xx[:, 1] = np.minimum(xx[:, 1], 300)          # rows 301..399 of field 1 are NEVER touched (pins F3)
    ys = [rng.integers(0, 2, size=(32, 1)).astype(np.int16) for _ in range(3)]
    gs = [rng.integers(0, 3, size=(32, 1)).astype(np.int64) for _ in range(3)]
    torch.manual_seed(SEED)
    three_steps("g3_ple3_adam", PLE(fd_sparse, D, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0),
                lambda m, xx, g: m(torch.from_numpy(xx)).gather(1, torch.from_numpy(g)).squeeze(1), xs, ys, gs)
    torch.manual_seed(SEED)
    three_steps("g3_dcn_adam", DCN(fd_sparse, D, 3, (32, 16, 8), dropout=0.0),
                lambda m, xx, g: m(torch.from_numpy(xx)), xs, ys, [None] * 3)

    # ---------------------------------------------------------------- G5 CDC modes
    def cdc_config(**kw):
        cfg = types.SimpleNamespace(mmoe_n_expert=4, ple_n_expert_specific=2, ple_n_expert_shared=2, gate_hidden_dim=8,
                                    dataset_name="golden", p_weight=0.5, p_weight_method="none", old_matrix_weight=0.0,
                                    affinity_func="minus", use_atten=False, n_cross_layers=3)
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg

    n_domain, n_cluster, domain_idx = 6, 3, 4
    fd_cdc = [7, 100, 3, 50, n_domain, 29]
    xc = _ids(rng, B, fd_cdc)
    d2g = np.array([0, 2, 1, 1, 0, 2], dtype=np.int64)
    for base, expert_dims, tower_dims in [("mmoe", (32, 16, 8), (8, 4)), ("ple", ((32, 16), (8,)), (8, 4)),
                                          ("star", (32, 16, 8), (32, 16, 8))]:
        torch.manual_seed(SEED)
        cdc = CDC(fd_cdc, D, n_cluster, n_domain, base, expert_dims, tower_dims, domain_idx,
                  domain_cnt_weight=np.full(n_domain, 1.0 / n_domain), n_causal_mask=4, dropout=0.0, config=cdc_config())
        cdc.domain2group = torch.from_numpy(d2g)
        cdc.domain2group_list = d2g.tolist()
        arrays = {"x": xc, "domain2group": d2g, "domain_idx": np.array(domain_idx)}
        arrays.update(_pack_sd("sd", cdc.state_dict()))
        cdc.eval()
        with torch.no_grad():
            arrays["eval_warmup"] = _np(cdc(torch.from_numpy(xc), mode="warmup"))
            arrays["eval_split"] = _np(cdc(torch.from_numpy(xc), mode="split"))
            arrays["eval_split_d3"] = _np(cdc(torch.from_numpy(xc), mode="split", domain_i=3))
        cdc.train()
        pred = cdc(torch.from_numpy(xc), mode="split")
        arrays["train_split"] = _np(pred)
        arrays["reg"] = _np(cdc.get_regularization_loss(device="cpu"))
        _save(f"g5_cdc_{base}", **arrays)

    # ---------------------------------------------------------------- G5b CDC(base='ple') TRAINED in split mode: the shape of
    # BASELINE config C4 (30 domains -> 4 clusters, emb_dim 32, nested expert dims: cdc.py:32-42) driven like run.py:635-640 —
    # two single-domain batches (mode='split', domain_i=d) and one mixed batch (domain_i=None), Adam on every parameter
    n_dom4, n_clu4, dom_idx4, D4, B4 = 30, 4, 2, 32, 48
    fd_c4 = [40, 900, n_dom4, 300, 9]
    d2g4 = np.array([(3 * d + 1) % n_clu4 for d in range(n_dom4)], dtype=np.int64)
    torch.manual_seed(SEED)
    cdc = CDC(fd_c4, D4, n_clu4, n_dom4, "ple", ((32, 16), (8,)), (8, 4), dom_idx4,
              domain_cnt_weight=np.full(n_dom4, 1.0 / n_dom4), n_causal_mask=4, dropout=0.0, config=cdc_config())
    cdc.domain2group = torch.from_numpy(d2g4)
    cdc.domain2group_list = d2g4.tolist()
    opt = torch.optim.Adam(params=cdc.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    arrays = {"domain2group": d2g4, "domain_idx": np.array(dom_idx4), "field_dims": np.array(fd_c4)}
    arrays.update(_pack_sd("sd0", cdc.state_dict()))
    cdc.train()
    for s_, dom in enumerate([7, 22, None]):
        xx = _ids(rng, B4, fd_c4)
        xx[:, 1] = np.minimum(xx[:, 1], 700)               # rows 701..899 of field 1 are never looked up (F3)
        if dom is not None:
            xx[:, dom_idx4] = dom                          # run.py:631: a batch of ONE domain
        yy = rng.integers(0, 2, size=(B4, 1)).astype(np.int16)
        pred = cdc(torch.from_numpy(xx), mode="split", domain_i=dom)
        loss = crit(pred.squeeze(), torch.from_numpy(yy).squeeze().float())
        bce = loss.detach().clone()
        reg = cdc.get_regularization_loss(device="cpu")
        loss = loss + reg
        cdc.zero_grad()
        loss.backward()
        opt.step()
        arrays[f"x{s_}"], arrays[f"y{s_}"] = xx, yy
        arrays[f"domain{s_}"] = np.array(-1 if dom is None else dom)
        arrays[f"bce{s_}"], arrays[f"reg{s_}"] = _np(bce), _np(reg)
        arrays.update(_pack_sd(f"sd{s_ + 1}", cdc.state_dict()))
    st = opt.state_dict()["state"]
    for i, (k, _) in enumerate(cdc.named_parameters()):
        if i in st:
            arrays[f"m3/{k}"] = _np(st[i]["exp_avg"])
            arrays[f"v3/{k}"] = _np(st[i]["exp_avg_sq"])
    _save("g5_cdc_ple_adam", **arrays)

    # ---------------------------------------------------------------- G6 batch of one
    x1 = x[:1].copy()
    arrays = {"x": x1, "x13": x13[:1].copy()}
    for name, ctor, xx in [
        ("ple", lambda: PLE(FD, D, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0), x1),
        ("mmoe", lambda: MMoE(FD, D, 3, 4, (32, 16, 8), (8, 4), dropout=0.0), x1),
        ("dcn", lambda: DCN(fd13, D, 3, (32, 16, 8), dropout=0.0), x13[:1]),
        ("star", lambda: STAR(FD, D, 3, (32, 16, 8), dropout=0.0), x1),
    ]:
        torch.manual_seed(SEED)
        m = ctor()
        arrays.update(_pack_sd(f"{name}/sd", m.state_dict()))
        m.train()
        arrays[f"{name}/train_pred"] = _np(m(torch.from_numpy(xx)))
        arrays.update(_pack_sd(f"{name}/sd_after", {k: v for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}))
    # CrossNetMix with B == 1: squeeze() drops the batch dim and the following cat raises (model/layer.py:406)
    torch.manual_seed(SEED)
    m = DCNv2(fd13, D, 2, (16, 8), dropout=0.0, low_rank=4, num_experts=2)
    try:
        m.train()
        m(torch.from_numpy(x13[:1]))
        arrays["dcnv2mix_b1_error"] = np.array("")
    except Exception as e:  # noqa: BLE001
        arrays["dcnv2mix_b1_error"] = np.array(type(e).__name__)
    _save("g6_batch1", **arrays)

    # ---------------------------------------------------------------- G7 BCE clamp on saturated probabilities
    logits = torch.tensor([-120.0, -40.0, -20.0, -5.0, 0.0, 5.0, 20.0, 40.0, 120.0] * 2)
    tgt = torch.tensor([0.0] * 9 + [1.0] * 9)
    p = torch.sigmoid(logits).requires_grad_(True)
    loss = crit(p, tgt)
    loss.backward()
    _save("g7_bce", p=_np(p), y=_np(tgt), loss=_np(loss), dp=_np(p.grad))

    # ---------------------------------------------------------------- G8 dropout statistics
    torch.manual_seed(SEED)
    m = PLE(FD, D, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.2)
    m.train()
    drop = torch.nn.Dropout(p=0.2)
    z = drop(torch.ones(200000))
    _save("g8_dropout", keep_rate=np.array(float((z != 0).float().mean())), kept_value=np.array(float(z.max())),
          p=np.array(0.2))

    # ---------------------------------------------------------------- G9 metrics from sklearn
    from sklearn.metrics import roc_auc_score, log_loss
    rng9 = np.random.default_rng(SEED + 9)
    n = 500
    t = rng9.integers(0, 2, size=n)
    s = np.round(rng9.random(n), 2).astype(np.float32)           # rounding creates ties
    dom = rng9.integers(0, 4, size=n)
    t[dom == 3] = 1                                               # single-class domain -> ValueError path (run.py:699-704)
    arrays = {"targets": t, "scores": s, "domains": dom, "auc": np.array(roc_auc_score(t, s)), "logloss": np.array(log_loss(t, s))}
    for d in range(4):
        mk = dom == d
        try:
            arrays[f"auc_d{d}"] = np.array(roc_auc_score(t[mk], s[mk]))
            arrays[f"logloss_d{d}"] = np.array(log_loss(t[mk], s[mk]))
        except ValueError:
            arrays[f"auc_d{d}"] = np.array(np.nan)
            arrays[f"logloss_d{d}"] = np.array(np.nan)
    _save("g9_metrics", **arrays)


if __name__ == "__main__":
    main()
