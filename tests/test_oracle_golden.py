"""Pins the CPU oracle (oracle/cdc_oracle.py) against the golden vectors captured from the reference's own
model/ package (tools/make_golden.py).  CPU only; this is the gate that lets the GPU tests trust the oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import O, assert_close, is_pre_bn_bias, oracle_grads

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# oracle and reference both run torch CPU fp32; differences come only from op grouping / summation order
RTOL, ATOL = 2e-5, 2e-6


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def sd_of(d, prefix="sd"):
    p = prefix + "/"
    return {k[len(p):]: torch.from_numpy(np.asarray(d[k])) for k in d.files if k.startswith(p)}


def test_g1_gather_bit_exact():
    d = load("g1_gather")
    fd = d["field_dims"].tolist()
    idx = O.gather_index(d["x"], fd)
    assert idx.dtype == np.int32 and np.array_equal(idx, d["idx"])
    out = O.embed(torch.from_numpy(d["table"]), d["x"], fd)
    assert torch.equal(out, torch.from_numpy(d["out"]))          # bit-exact row selection
    assert torch.equal(out.view(64, len(fd), -1), torch.from_numpy(d["out3d"]))


def test_gather_index_wraps_like_int32():
    fd = [2 ** 30, 2 ** 30, 2 ** 30]            # offsets 0, 2^30, 2^31 -> wraps negative in int32
    x = np.array([[1, 2, 3]], dtype=np.int32)
    want = (torch.from_numpy(x) + torch.from_numpy(x).new_tensor(O.field_offsets(fd))).numpy()
    assert np.array_equal(O.gather_index(x, fd), want)


MODELS = {
    "g2_ple3": ("ple", lambda sd, x, tr, so: O.ple_forward(sd, x, FD, 3, tr, so)),
    "g2_mmoe4": ("mmoe", lambda sd, x, tr, so: O.mmoe_forward(sd, x, FD, 3, tr, so)),
    "g2_mmoe8": ("mmoe", lambda sd, x, tr, so: O.mmoe_forward(sd, x, FD, 3, tr, so)),
    "g2_dcn13": ("dcn", lambda sd, x, tr, so: O.dcn_forward(sd, x, FD13, tr, so)),
    "g2_dcnv2_mix": ("dcnv2", lambda sd, x, tr, so: O.dcnv2_forward(sd, x, FD13, tr, so)),
    "g2_dcnv2_stacked": ("dcnv2", lambda sd, x, tr, so: O.dcnv2_forward(sd, x, FD13, tr, so, model_structure="stacked")),
    "g2_star5_all": ("star", lambda sd, x, tr, so: O.star_forward(sd, x, FD, 5, training=tr, stats_out=so)),
    "g2_star30_all": ("star", lambda sd, x, tr, so: O.star_forward(sd, x, FD, 30, training=tr, stats_out=so)),
    "g12_ple3_atten": ("ple", lambda sd, x, tr, so: O.ple_forward(sd, x, FD, 3, tr, so)),
    "g12_mmoe4_atten_nores": ("mmoe", lambda sd, x, tr, so: O.mmoe_forward(sd, x, FD, 3, tr, so)),
    "g12_star3_atten": ("star", lambda sd, x, tr, so: O.star_forward(sd, x, FD, 3, training=tr, stats_out=so)),
    "g13_autoint": ("autoint", lambda sd, x, tr, so: O.autoint_forward(sd, x, FD, tr, so)),
    "g14_hinet": ("hinet", lambda sd, x, tr, so: O.hinet_forward(sd, x, FD, load("g14_hinet")["group"], 2, tr, so)),
    "g13_adasparse": ("adasparse", lambda sd, x, tr, so: O.adasparse_forward(sd, x, FD, 2, tr, so)),
    "g16_pepnet": ("pepnet", lambda sd, x, tr, so: O.pepnet_forward(sd, x, FD, 2, 3, tr, so)),
    "g16_epnet": ("pepnet", lambda sd, x, tr, so: O.pepnet_forward(sd, x, FD, 2, 3, tr, so)),
    "g13_epnet_single": ("pepnet", lambda sd, x, tr, so: O.pepnet_forward(sd, x, FD, 2, 1, tr, so)),
    "g11_deepfm": ("deepfm", lambda sd, x, tr, so: O.deepfm_forward(sd, x, FD13, tr, so)),
}
FD = [7, 100, 3, 50, 11, 29]
FD13 = [11, 50, 7, 100, 3, 29, 64, 5, 17, 200, 9, 31, 13]


def _l2_map(sd, kind):
    return {n: 1e-5 for n in O.reg_names(list(sd), kind)}


@pytest.mark.parametrize("name", sorted(MODELS))
def test_g2_model_forward_backward(name):
    kind, fwd = MODELS[name]
    d = load(name)
    sd = sd_of(d)
    x, y = d["x"], torch.from_numpy(d["y"]).reshape(-1)
    group = torch.from_numpy(d["group"]) if "group" in d.files and name != "g14_hinet" else None    # HiNet reads the group itself
    stats = {}

    def loss_fn(s):
        p = fwd(s, x, True, stats)
        p = p.gather(1, group).squeeze(1) if group is not None else p
        loss_fn.pred = p
        loss_fn.bce = O.bce_mean(p, y)
        loss_fn.reg = O.reg_loss(s, _l2_map(s, kind))
        return (loss_fn.bce + loss_fn.reg).sum()

    _, grads = oracle_grads(loss_fn, sd, torch.tensor(1.0))
    assert_close(loss_fn.pred, d["train_pred"], RTOL, ATOL, "train_pred")
    assert_close(loss_fn.bce, d["bce"], RTOL, ATOL, "bce")
    assert_close(loss_fn.reg, d["reg"], RTOL, ATOL, "reg")
    n_checked = 0
    for k in d.files:
        if k.startswith("grad/"):
            g = grads[k[5:]]
            assert g is not None, k
            if is_pre_bn_bias(k[5:], set(sd)) and float(np.abs(d[k]).max()) < 1e-4:
                assert float(g.abs().max()) < 1e-4, k       # rounding noise on both sides
                continue                                    # (a gated Linear -> BatchNorm, as in AdaSparse, has a real bias gradient)
            scale = max(float(np.abs(d[k]).max()), 1.0)
            assert_close(g, d[k], 1e-4, 1e-6 * scale + 2e-7, k)
            n_checked += 1
    assert n_checked > 5
    # parameters the reference leaves without a gradient must have none / zero here too
    for k, g in grads.items():
        if "grad/" + k not in d.files and g is not None:
            assert float(g.abs().max()) == 0.0, f"{k} has a gradient the reference does not produce"
    for k in d.files:
        if k.startswith("sd_after/"):
            assert_close(stats[k[9:]], d[k], RTOL, ATOL, k) if k[9:] in stats else None
    sd_eval = dict(sd)
    sd_eval.update(stats)                  # the reference's eval forward ran after the training one
    ev = fwd(sd_eval, x, False, None)
    ev = ev.gather(1, group).squeeze(1) if group is not None else ev
    assert_close(ev, d["eval_pred"], RTOL, ATOL, "eval_pred")


def test_g2_crossnetv2_layer():
    d = load("g2_crossnetv2_layer")
    sd = {"cn." + k: v for k, v in sd_of(d).items()}
    x = torch.from_numpy(d["x"]).requires_grad_(True)
    for v in sd.values():
        v.requires_grad_(True)
    out = O.cross_network_v2(x, sd, "cn")
    out.backward(torch.from_numpy(d["gout"]))
    assert_close(out, d["out"], RTOL, ATOL, "out")
    assert_close(x.grad, d["dx"], 1e-4, 1e-5, "dx")
    for k in d.files:
        if k.startswith("grad/"):
            assert_close(sd["cn." + k[5:]].grad, d[k], 1e-4, 1e-5, k)


def test_g2_dcnv2_constructor_errors_recorded():
    d = load("g2_dcnv2_ctor_errors")
    assert str(d["v2"]) == "AttributeError" and str(d["crossnet_only"]) == "AttributeError"
    assert str(d["bad_structure"]) == "AssertionError"


def test_g4_star_grouped():
    d = load("g4_star5_grouped")
    sd = sd_of(d)
    group = d["group"]
    assert (group == 3).sum() == 0 and (group == 4).sum() == 1      # an empty group and a one-row group
    y = torch.from_numpy(d["y"])
    stats = {}

    def loss_fn(s):
        p, t = O.star_forward(s, d["x"], FD, 5, x_group=group, targets=y, training=True, stats_out=stats)
        loss_fn.p, loss_fn.t = p, t
        return (O.bce_mean(p.squeeze(), t.squeeze()) + O.reg_loss(s, _l2_map(s, "star"))).sum()

    _, grads = oracle_grads(loss_fn, sd, torch.tensor(1.0))
    assert_close(loss_fn.p, d["train_pred"], RTOL, ATOL, "train_pred")
    assert torch.equal(loss_fn.t, torch.from_numpy(d["train_targets"]))
    for k in d.files:
        if k.startswith("grad/"):
            if is_pre_bn_bias(k[5:], set(sd)):
                continue
            scale = max(float(np.abs(d[k]).max()), 1.0)
            g = grads[k[5:]]
            if g is None:                      # the empty group: the reference reports exact zeros
                assert not np.any(d[k]), k
                continue
            assert_close(g, d[k], 1e-4, 1e-6 * scale + 2e-7, k)
    for k in d.files:
        if k.startswith("sd_after/") and k[9:] in stats:
            assert_close(stats[k[9:]], d[k], RTOL, ATOL, k)
    # the empty group's statistics are untouched by the reference
    assert np.array_equal(d["sd_after/domain_norm.3.running_mean"], d["sd/domain_norm.3.running_mean"])
    sd_eval = dict(sd)
    sd_eval.update(stats)
    pe, te = O.star_forward(sd_eval, d["x"], FD, 5, x_group=group, targets=y, training=False)
    assert_close(pe, d["eval_pred"], RTOL, ATOL, "eval_pred")
    assert torch.equal(te, torch.from_numpy(d["eval_targets"]))


@pytest.mark.parametrize("name,kind", [("g3_ple3_adam", "ple"), ("g3_dcn_adam", "dcn")])
def test_g3_three_adam_steps(name, kind):
    d = load(name)
    fd = [7, 400, 3, 50, 11, 29]
    sd = sd_of(d, "sd0")
    m = {k: torch.zeros_like(v) for k, v in sd.items() if v.dtype.is_floating_point}
    v = {k: torch.zeros_like(t) for k, t in m.items()}
    l2 = None
    for s in range(3):
        x, y = d[f"x{s}"], torch.from_numpy(d[f"y{s}"]).reshape(-1)
        group = torch.from_numpy(d[f"group{s}"]) if f"group{s}" in d.files else None
        stats = {}

        def loss_fn(sdd):
            nonlocal l2
            if kind == "ple":
                p = O.ple_forward(sdd, x, fd, 3, True, stats).gather(1, group).squeeze(1)
            else:
                p = O.dcn_forward(sdd, x, fd, True, stats)
            l2 = _l2_map(sdd, kind)
            loss_fn.bce = O.bce_mean(p, y)
            loss_fn.reg = O.reg_loss(sdd, l2)
            return (loss_fn.bce + loss_fn.reg).sum()

        _, grads = oracle_grads(loss_fn, sd, torch.tensor(1.0))
        assert_close(loss_fn.bce, d[f"bce{s}"], 1e-4, 1e-6, f"bce{s}")
        assert_close(loss_fn.reg, d[f"reg{s}"], 1e-5, 1e-7, f"reg{s}")
        new = dict(sd)
        new.update(stats)
        for k, g in grads.items():
            if g is None:
                continue
            new[k], m[k], v[k] = O.adam_step(sd[k], g, m[k], v[k], s + 1)
        sd = new
        for k in sd:
            gk = f"sd{s + 1}/{k}"
            if is_pre_bn_bias(k, set(sd)):
                # gradient = rounding noise, and Adam turns its SIGN into a +-lr move: not reproducible by any
                # other summation order (and without effect on the outputs: the batch mean removes the bias).
                # Adopt the reference's value so the BatchNorm running means that contain it stay comparable.
                sd[k] = torch.from_numpy(d[gk])
                m[k] = torch.from_numpy(d[f"m{s + 1}/{k}"])
                v[k] = torch.from_numpy(d[f"v{s + 1}/{k}"])
                continue
            assert_close(sd[k], d[gk], 2e-5, 2e-6, gk)
        for k in m:
            if is_pre_bn_bias(k, set(sd)):
                continue
            if f"m{s + 1}/{k}" in d.files:
                assert_close(m[k], d[f"m{s + 1}/{k}"], 5e-4, 1e-7, f"m{s + 1}/{k}")
                assert_close(v[k], d[f"v{s + 1}/{k}"], 1e-3, 1e-10, f"v{s + 1}/{k}")
    # F3: rows 301..399 of field 1 are never looked up, yet the reference moved them by ~lr per step
    off = 7
    w0 = d["sd0/embedding.embedding_dict.weight"][off + 301:off + 400]
    w3 = d["sd3/embedding.embedding_dict.weight"][off + 301:off + 400]
    moved = np.abs(w3 - w0)
    assert moved.min() > 2.5e-3 and moved.max() < 3.5e-3


@pytest.mark.parametrize("base", ["mmoe", "ple", "star"])
def test_g5_cdc_modes(base):
    d = load(f"g5_cdc_{base}")
    fd = [7, 100, 3, 50, 6, 29]
    sd = sd_of(d)
    pre = "base_model_instance."
    fwd = {"mmoe": lambda x, tr: O.mmoe_forward(sd, x, fd, 3, tr, None, prefix=pre),
           "ple": lambda x, tr: O.ple_forward(sd, x, fd, 3, tr, None, prefix=pre),
           "star": lambda x, tr: O.star_forward(sd, x, fd, 3, training=tr, prefix=pre)}[base]
    x, d2g, di = d["x"], d["domain2group"], int(d["domain_idx"])
    assert_close(O.cdc_forward(lambda xx: fwd(xx, False), x, d2g, di, "warmup"), d["eval_warmup"], RTOL, ATOL, "warmup")
    assert_close(O.cdc_forward(lambda xx: fwd(xx, False), x, d2g, di, "split"), d["eval_split"], RTOL, ATOL, "split")
    assert_close(O.cdc_forward(lambda xx: fwd(xx, False), x, d2g, di, "split", domain_i=3), d["eval_split_d3"], RTOL, ATOL, "d3")
    assert_close(O.cdc_forward(lambda xx: fwd(xx, True), x, d2g, di, "split"), d["train_split"], RTOL, ATOL, "train split")
    kind = base
    l2 = {n: 1e-5 for n in O.reg_names(list(sd), kind)}
    assert_close(O.reg_loss(sd, l2), d["reg"], RTOL, ATOL, "reg")


def test_g6_batch_of_one():
    d = load("g6_batch1")
    x1, x13 = d["x"], d["x13"]
    for name, f in [("ple", lambda sd, so: O.ple_forward(sd, x1, FD, 3, True, so)),
                    ("mmoe", lambda sd, so: O.mmoe_forward(sd, x1, FD, 3, True, so)),
                    ("dcn", lambda sd, so: O.dcn_forward(sd, x13, FD13, True, so)),
                    ("star", lambda sd, so: O.star_forward(sd, x1, FD, 3, training=True, stats_out=so))]:
        sd = sd_of(d, f"{name}/sd")
        stats = {}
        assert_close(f(sd, stats), d[f"{name}/train_pred"], RTOL, ATOL, name)
        # BatchNorm is skipped for one row: no statistic moves (model/layer.py:202, star.py:94,134)
        assert not stats, f"{name}: statistics changed for a batch of one"
        for k in d.files:
            if k.startswith(f"{name}/sd_after/"):
                assert np.array_equal(d[k], d[k.replace("sd_after", "sd")]), k
    assert str(d["dcnv2mix_b1_error"]) == "IndexError"


def test_g7_bce_clamp():
    d = load("g7_bce")
    p = torch.from_numpy(d["p"]).requires_grad_(True)
    y = torch.from_numpy(d["y"])
    loss = O.bce_mean(p, y)
    assert_close(loss, d["loss"], 1e-6, 1e-6, "loss")
    # gradient as aten::binary_cross_entropy_backward: (p - y) / max((1-p)*p, 1e-12) / N
    dp = (p.detach() - y) / torch.clamp((1 - p.detach()) * p.detach(), min=1e-12) / p.numel()
    assert_close(dp, d["dp"], 1e-6, 0.0, "dp")


def test_g9_metrics_match_sklearn():
    d = load("g9_metrics")
    t, s, dom = d["targets"], d["scores"], d["domains"]
    assert abs(O.auc(t, s) - float(d["auc"])) < 1e-12
    assert abs(O.logloss(t, s) - float(d["logloss"])) < 1e-13
    for k in range(4):
        mk = dom == k
        if np.isnan(d[f"auc_d{k}"]):
            with pytest.raises(ValueError):
                O.auc(t[mk], s[mk])
        else:
            assert abs(O.auc(t[mk], s[mk]) - float(d[f"auc_d{k}"])) < 1e-12
            assert abs(O.logloss(t[mk], s[mk]) - float(d[f"logloss_d{k}"])) < 1e-13


def test_g17_adl_forward_backward():
    """ADL: routing + per-tower MLPs on their own rows, centres after each call (tests/golden/g17_adl.npz)."""
    d = load("g17_adl")
    sd = sd_of(d)
    x, y = d["x"], torch.from_numpy(d["y"])
    c0 = torch.from_numpy(d["centers0"])
    stats, keep = {}, {}

    def loss_fn(s):
        p, t, keep["c1"] = O.adl_forward(s, x, FD, c0, 3, targets=y, is_training=True, training=True, stats_out=stats)
        keep["pred"], keep["t"] = p, t
        keep["bce"] = O.bce_mean(p.squeeze(1), t.reshape(-1).float())
        keep["reg"] = O.reg_loss(s, _l2_map(s, "adl"))
        return (keep["bce"] + keep["reg"]).sum()

    _, grads = oracle_grads(loss_fn, sd, torch.tensor(1.0))
    assert torch.equal(keep["t"], torch.from_numpy(d["train_targets"]))
    assert_close(keep["pred"], d["train_pred"], RTOL, ATOL, "train_pred")
    assert_close(keep["bce"], d["bce"], RTOL, ATOL, "bce")
    assert_close(keep["reg"], d["reg"], RTOL, ATOL, "reg")
    assert_close(keep["c1"], d["centers1"], 1e-6, 1e-7, "centres")
    for k in d.files:
        if k.startswith("grad/") and not is_pre_bn_bias(k[5:], set(sd)):
            assert_close(grads[k[5:]], d[k], 1e-4, 1e-6 * max(float(np.abs(d[k]).max()), 1.0) + 2e-7, k)
    for k in d.files:
        if k.startswith("sd_after/") and "num_batches" not in k:
            if k[9:] in stats:
                assert_close(stats[k[9:]], d[k], RTOL, ATOL, k)
            else:                                           # shared_mlps: constructed, never called (model/adl.py:92-93)
                assert_close(sd[k[9:]], d[k], 0, 0, k)
    sd_eval = dict(sd)
    sd_eval.update(stats)
    ev, _, c2 = O.adl_forward(sd_eval, x, FD, keep["c1"], 3, is_training=False, training=False)
    assert_close(ev, d["eval_pred"], RTOL, ATOL, "eval_pred")
    assert_close(c2, d["centers2"], 1e-6, 1e-7, "centres after eval")
