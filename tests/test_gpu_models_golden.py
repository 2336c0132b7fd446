"""Every model of the hot path, driven exactly like the reference trainer drives it (run.py:481-492: model(X) ->
BCELoss -> + get_regularization_loss -> backward), against the golden vectors captured from the reference itself
(tools/make_golden.py): predictions (train / eval), BCE, regularisation term, every parameter gradient incl. the
dense table gradient, BatchNorm statistics.  fp32 MFMA path; tolerances state what differs (summation order only)."""
import os
import types

import numpy as np
import pytest
import torch

from helpers import assert_close, is_pre_bn_bias

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FD = [7, 100, 3, 50, 11, 29]
FD13 = [11, 50, 7, 100, 3, 29, 64, 5, 17, 200, 9, 31, 13]
RTOL, ATOL = 2e-5, 2e-6        # 3.6x the worst share measured over 2 239 comparisons (profiles/round3/parity_margins.txt); SURVEY 8c proposed 1e-5 / 1e-6


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def sd_of(d, prefix="sd"):
    p = prefix + "/"
    return {k[len(p):]: torch.from_numpy(np.asarray(d[k])) for k in d.files if k.startswith(p)}


def build(name):
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.model.dcn import DCN
    from cdcmdr_amd.model.dcnv2 import DCNv2
    from cdcmdr_amd.model.star import STAR
    return {
        "g2_ple3": lambda: PLE(FD, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0),
        "g2_mmoe4": lambda: MMoE(FD, 4, 3, 4, (32, 16, 8), (8, 4), dropout=0.0),
        "g2_mmoe8": lambda: MMoE(FD, 4, 3, 8, (32, 16, 8), (8, 4), dropout=0.0),
        "g2_dcn13": lambda: DCN(FD13, 4, 3, (32, 16, 8), dropout=0.0),
        "g2_dcnv2_mix": lambda: DCNv2(FD13, 4, 3, (32, 16, 8), dropout=0.0, low_rank=8, num_experts=4),
        "g2_dcnv2_stacked": lambda: DCNv2(FD13, 4, 2, (32, 16), dropout=0.0, model_structure="stacked", low_rank=8),
        "g2_star5_all": lambda: STAR(FD, 4, 5, (32, 16, 8), dropout=0.0),
        "g2_star30_all": lambda: STAR(FD, 4, 30, (16, 8), dropout=0.0),
        "g12_ple3_atten": lambda: PLE(FD, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), 0.0, _atten_cfg(True)),
        "g12_mmoe4_atten_nores": lambda: MMoE(FD, 4, 3, 4, (32, 16, 8), (8, 4), 0.0, _atten_cfg(False)),
        "g12_star3_atten": lambda: STAR(FD, 4, 3, (32, 16, 8), None, 0.0, _atten_cfg(True)),
        "g13_autoint": lambda: __import__("cdcmdr_amd.model.autoint", fromlist=["AutoInt"]).AutoInt(
            FD, 4, atten_embed_dim=8, att_layer_num=2, att_head_num=2, att_res=True, mlp_dims=(32, 16), dropout=0.0),
        "g13_adasparse": lambda: __import__("cdcmdr_amd.model.adasparse", fromlist=["AdaSparse"]).AdaSparse(
            FD, 4, (32, 16, 8), domain_idx=2, dropout=0.0, config=_atten_cfg_off()),
        "g16_pepnet": lambda: _pepnet(3, True),
        "g16_epnet": lambda: _pepnet(3, False),
        "g13_epnet_single": lambda: _pepnet(1, False),
        "g11_deepfm": lambda: __import__("cdcmdr_amd.model.dfm", fromlist=["DeepFM"]).DeepFM(FD13, 4, (32, 16, 8), dropout=0.0),
    }[name]()


def _pepnet(n_tower, use_ppnet):
    from cdcmdr_amd.model.pepnet import PEPNet
    return PEPNet(FD, 4, n_tower, (16, 8), gate_hidden_dim=8, domain_idx=2, use_ppnet=use_ppnet, dropout=0.0, config=_atten_cfg_off())


def _atten_cfg_off():
    import types
    return types.SimpleNamespace(use_atten=False, use_dcn=False)


def _atten_cfg(att_res):
    import types
    return types.SimpleNamespace(use_atten=True, atten_embed_dim=8, att_layer_num=2, att_head_num=2, att_res=att_res, use_dcn=False)


def check_grads(model, d, names):
    params = dict(model.named_parameters())
    n = 0
    for k in d.files:
        if not k.startswith("grad/"):
            continue
        name = k[5:]
        got = params[name].grad
        assert got is not None, f"no gradient for {name}"
        want = d[k]
        if is_pre_bn_bias(name, names) and float(got.abs().max()) < 1e-4 and float(np.abs(want).max()) < 1e-4:
            continue                       # rounding noise on both sides (a one-row group skips BatchNorm: compared below)
        scale = max(float(np.abs(want).max()), 1.0)
        assert_close(got, want, RTOL, ATOL * scale, k)
        n += 1
    assert n > 5
    for name, p in params.items():
        if "grad/" + name not in d.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{name}: gradient the reference does not produce"


@pytest.mark.parametrize("name", ["g2_ple3", "g2_mmoe4", "g2_mmoe8", "g2_dcn13", "g2_dcnv2_mix", "g2_dcnv2_stacked",
                                  "g2_star5_all", "g2_star30_all", "g11_deepfm", "g12_ple3_atten", "g12_mmoe4_atten_nores",
                                  "g12_star3_atten", "g13_autoint", "g13_adasparse", "g16_pepnet", "g16_epnet", "g13_epnet_single"])
def test_model_matches_reference_golden(cuda, name):
    d = load(name)
    model = build(name).to(cuda).set_precision("f32")
    model.load_state_dict(sd_of(d))
    names = set(model.state_dict().keys())
    x = torch.from_numpy(d["x"]).to(cuda)
    y = torch.from_numpy(d["y"]).to(cuda).reshape(-1).float()
    group = torch.from_numpy(d["group"]).to(cuda) if "group" in d.files else None
    crit = torch.nn.BCELoss()
    model.train()
    pred = model(x)
    pred = pred.gather(1, group).squeeze(1) if group is not None else pred
    bce = crit(pred, y)
    reg = model.get_regularization_loss(device=cuda)
    loss = bce + reg
    model.zero_grad()
    loss.backward()
    assert_close(pred, d["train_pred"], RTOL, ATOL, "train_pred")
    assert_close(bce, d["bce"], RTOL, ATOL, "bce")
    assert_close(reg.reshape(-1), d["reg"].reshape(-1), 1e-5, 1e-7, "reg")
    check_grads(model, d, names)
    sd = model.state_dict()
    for k in d.files:
        if k.startswith("sd_after/"):
            assert_close(sd[k[9:]], d[k], RTOL, ATOL, k)
    model.eval()
    with torch.no_grad():
        ev = model(x)
        ev = ev.gather(1, group).squeeze(1) if group is not None else ev
    assert_close(ev, d["eval_pred"], RTOL, ATOL, "eval_pred")


def test_dcnv2_constructor_errors_like_reference(cuda):
    from cdcmdr_amd.model.dcnv2 import DCNv2
    d = load("g2_dcnv2_ctor_errors")
    for tag, kw in [("v2", dict(use_low_rank_mixture=False)), ("crossnet_only", dict(model_structure="crossnet_only")),
                    ("bad_structure", dict(model_structure="nope"))]:
        with pytest.raises(Exception) as e:
            DCNv2(FD13, 4, 2, (32, 16), dropout=0.0, **kw)
        assert type(e.value).__name__ == str(d[tag])


def test_crossnetv2_layer_golden(cuda):
    from cdcmdr_amd.model.layer import CrossNetV2
    d = load("g2_crossnetv2_layer")
    cn = CrossNetV2(24, 3).to(cuda).set_precision("f32")
    cn.load_state_dict(sd_of(d))
    x = torch.from_numpy(d["x"]).to(cuda).requires_grad_(True)
    out = cn(x)
    out.backward(torch.from_numpy(d["gout"]).to(cuda))
    assert_close(out, d["out"], RTOL, ATOL, "out")
    assert_close(x.grad, d["dx"], RTOL, 1e-4, "dx")
    for k, p in cn.named_parameters():
        assert_close(p.grad, d[f"grad/{k}"], RTOL, 1e-4, k)


def test_star_grouped_mode_golden(cuda):
    """G4: rows re-ordered by group, targets permuted the same way, a one-row group (BN skipped) and an empty group."""
    from cdcmdr_amd.model.star import STAR
    d = load("g4_star5_grouped")
    model = STAR(FD, 4, 5, (32, 16, 8), dropout=0.0).to(cuda).set_precision("f32")
    model.load_state_dict(sd_of(d))
    names = set(model.state_dict().keys())
    x = torch.from_numpy(d["x"]).to(cuda)
    y = torch.from_numpy(d["y"]).to(cuda)
    g = torch.from_numpy(d["group"]).to(cuda)
    model.train()
    pred, yy = model(x, g, targets=y)
    assert torch.equal(yy.cpu(), torch.from_numpy(d["train_targets"]))          # stable partition, bit-exact
    loss = torch.nn.BCELoss()(pred.squeeze(), yy.squeeze().float()) + model.get_regularization_loss(device=cuda)
    model.zero_grad()
    loss.backward()
    assert_close(pred, d["train_pred"], RTOL, ATOL, "train_pred")
    check_grads(model, d, names)
    sd = model.state_dict()
    for k in d.files:
        if k.startswith("sd_after/"):
            assert_close(sd[k[9:]], d[k], RTOL, ATOL, k)
    model.eval()
    with torch.no_grad():
        pe, ye = model(x, g, targets=y)
    assert_close(pe, d["eval_pred"], RTOL, ATOL, "eval_pred")
    assert torch.equal(ye.cpu(), torch.from_numpy(d["eval_targets"]))


@pytest.mark.parametrize("base,expert_dims,tower_dims", [("mmoe", (32, 16, 8), (8, 4)), ("ple", ((32, 16), (8,)), (8, 4)),
                                                         ("star", (32, 16, 8), (32, 16, 8))])
def test_cdc_modes_golden(cuda, base, expert_dims, tower_dims):
    from cdcmdr_amd.model.cdc import CDC
    d = load(f"g5_cdc_{base}")
    cfg = types.SimpleNamespace(mmoe_n_expert=4, ple_n_expert_specific=2, ple_n_expert_shared=2, gate_hidden_dim=8,
                                dataset_name="golden", p_weight=0.5, p_weight_method="none", old_matrix_weight=0.0,
                                affinity_func="minus", use_atten=False, n_cross_layers=3)
    fd = [7, 100, 3, 50, 6, 29]
    cdc = CDC(fd, 4, 3, 6, base, expert_dims, tower_dims, int(d["domain_idx"]), domain_cnt_weight=np.full(6, 1 / 6),
              n_causal_mask=4, dropout=0.0, config=cfg, device=cuda).to(cuda).set_precision("f32")
    cdc.load_state_dict(sd_of(d))
    cdc.domain2group = torch.from_numpy(d["domain2group"]).to(cuda)
    cdc.domain2group_list = d["domain2group"].tolist()
    x = torch.from_numpy(d["x"]).to(cuda)
    cdc.eval()
    with torch.no_grad():
        assert_close(cdc(x, mode="warmup"), d["eval_warmup"], RTOL, ATOL, "warmup")
        assert_close(cdc(x, mode="split"), d["eval_split"], RTOL, ATOL, "split")
        assert_close(cdc(x, mode="split", domain_i=3), d["eval_split_d3"], RTOL, ATOL, "split d3")
    cdc.train()
    assert_close(cdc(x, mode="split"), d["train_split"], RTOL, ATOL, "train split")
    assert_close(cdc.get_regularization_loss(cuda).reshape(-1), d["reg"].reshape(-1), 1e-5, 1e-7, "reg")
    # in-memory snapshot / rollback of the base model (cdc.py:343-354)
    cdc.save_model_state()
    w = cdc.base_model_instance.linear.fc.weight
    before = w.detach().clone()
    w.data.add_(1.0)
    cdc.load_model_state()
    assert torch.equal(w.detach(), before)


def test_batch_of_one_golden(cuda):
    """G6: BatchNorm is skipped for a single row and no statistic moves; CrossNetMix raises like the reference."""
    d = load("g6_batch1")
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.model.dcn import DCN
    from cdcmdr_amd.model.dcnv2 import DCNv2
    from cdcmdr_amd.model.star import STAR
    ctors = {"ple": (lambda: PLE(FD, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0), "x"),
             "mmoe": (lambda: MMoE(FD, 4, 3, 4, (32, 16, 8), (8, 4), dropout=0.0), "x"),
             "dcn": (lambda: DCN(FD13, 4, 3, (32, 16, 8), dropout=0.0), "x13"),
             "star": (lambda: STAR(FD, 4, 3, (32, 16, 8), dropout=0.0), "x")}
    for name, (ctor, xk) in ctors.items():
        m = ctor().to(cuda).set_precision("f32")
        m.load_state_dict(sd_of(d, f"{name}/sd"))
        m.train()
        with torch.no_grad():
            out = m(torch.from_numpy(d[xk]).to(cuda))
        assert_close(out, d[f"{name}/train_pred"], RTOL, ATOL, name)
        sd = m.state_dict()
        for k in d.files:
            if k.startswith(f"{name}/sd_after/"):
                assert torch.equal(sd[k[len(name) + 10:]].cpu(), torch.from_numpy(d[k])), k
    m = DCNv2(FD13, 4, 2, (16, 8), dropout=0.0, low_rank=4, num_experts=2).to(cuda)
    m.train()
    with pytest.raises(Exception) as e:
        m(torch.from_numpy(d["x13"]).to(cuda))
    assert type(e.value).__name__ == str(d["dcnv2mix_b1_error"])


def test_fm_term_golden(cuda):
    """cdc_fm_fwd / cdc_fm_bwd against the reference's FactorizationMachine(reduce_sum=True) (model/layer.py:160-175)."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    d = load("g11_deepfm")
    e = torch.from_numpy(d["fm_in"]).to(cuda)
    B, F, D = e.shape
    e2 = e.reshape(B, F * D).contiguous()
    out = torch.empty((B, 1), dtype=torch.float32, device=cuda)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.cdc_fm_fwd(e2.data_ptr(), F * D, out.data_ptr(), 1, B, F, D, s), "fm_fwd")
    assert_close(out, d["fm_out"], 1e-5, 1e-6, "fm_out")
    de = torch.full((B, F * D), 7.0, dtype=torch.float32, device=cuda)
    ones = torch.ones((B, 1), dtype=torch.float32, device=cuda)
    L.check(lib.cdc_fm_bwd(e2.data_ptr(), F * D, ones.data_ptr(), 1, de.data_ptr(), F * D, B, F, D, 0, s), "fm_bwd")
    assert_close(de.reshape(B, F, D), d["fm_grad"], 1e-5, 1e-6, "fm_grad")
    L.check(lib.cdc_fm_bwd(e2.data_ptr(), F * D, ones.data_ptr(), 1, de.data_ptr(), F * D, B, F, D, 1, s), "fm_bwd(acc)")
    assert_close(de.reshape(B, F, D), 2 * d["fm_grad"], 1e-5, 1e-6, "fm_grad accumulated")


@pytest.mark.parametrize("B,F,A,H", [(5, 6, 8, 2), (64, 26, 64, 2), (3, 64, 16, 4), (7, 1, 4, 1)])
def test_attention_core_against_torch(cuda, B, F, A, H):
    """cdc_attn_fwd / cdc_attn_bwd (the inside of nn.MultiheadAttention, model/layer.py:75-77) against torch's own
    scaled_dot_product_attention on the CPU in fp32 — probabilities, output, and the gradient of all three projections."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    torch.manual_seed(B * 100 + F)
    qkv = torch.randn(B * F, 3 * A)
    dout = torch.randn(B * F, A)
    ref = qkv.clone().requires_grad_(True)
    dh = A // H
    q, k, v = [t.reshape(B, F, H, dh).transpose(1, 2) for t in ref.split(A, dim=1)]            # [B, H, F, dh]
    pr = torch.softmax((q * dh ** -0.5) @ k.transpose(-1, -2), dim=-1)
    want = (pr @ v).transpose(1, 2).reshape(B * F, A)
    want.backward(dout)
    d_qkv = qkv.to(cuda).contiguous()
    out = torch.empty((B * F, A), device=cuda)
    probs = torch.empty((B, H, F, F), device=cuda)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.cdc_attn_fwd(d_qkv.data_ptr(), 3 * A, out.data_ptr(), A, probs.data_ptr(), B, F, A, H, 0.0, 1, None, s), "attn_fwd")
    assert_close(probs, pr.detach(), 1e-5, 1e-6, "probs")
    assert_close(out, want.detach(), 1e-5, 1e-6, "attention output")
    dq = torch.full((B * F, 3 * A), 9.0, device=cuda)
    L.check(lib.cdc_attn_bwd(d_qkv.data_ptr(), 3 * A, probs.data_ptr(), dout.to(cuda).data_ptr(), A, dq.data_ptr(), 3 * A, B, F, A, H,
                             0.0, 1, None, s), "attn_bwd")
    assert_close(dq, ref.grad, 2e-5, 2e-6, "d qkv")
    # dropout on the probabilities: the backward regenerates the forward's decisions (finite-difference-free check:
    # with dout = v-independent ones, d v equals the column sums of the dropped probabilities the forward used)
    out2 = torch.empty_like(out)
    L.check(lib.cdc_attn_fwd(d_qkv.data_ptr(), 3 * A, out2.data_ptr(), A, probs.data_ptr(), B, F, A, H, 0.5, 7, None, s), "attn_fwd(drop)")
    if F > 1:
        assert float((out2 - out).abs().max()) > 0
    ones = torch.ones((B * F, A), device=cuda)
    L.check(lib.cdc_attn_bwd(d_qkv.data_ptr(), 3 * A, probs.data_ptr(), ones.data_ptr(), A, dq.data_ptr(), 3 * A, B, F, A, H, 0.5, 7, None, s),
            "attn_bwd(drop)")
    dv = dq[:, 2 * A:].reshape(B, F, H, dh)                       # sum_i Pdrop[i, j] for every d
    vsum = d_qkv[:, 2 * A:].reshape(B, F, H, dh)
    # out2[b, i, h, :] = sum_j Pdrop[i, j] v[j]  =>  sum_i out2 = sum_j (sum_i Pdrop[i, j]) v[j] = sum_j dv[j, 0] * v[j]
    lhs = out2.reshape(B, F, H, dh).sum(dim=1)
    rhs = (dv[..., :1] * vsum).sum(dim=1)
    assert_close(lhs, rhs, 1e-4, 1e-5, "dropout mask consistent between forward and backward")


def test_hinet_matches_reference_golden(cuda):
    """G14: HiNet(x, x_group, targets) -> (pred, targets): forward, BCE, regularisation term, every gradient, BatchNorm
    statistics, eval forward against the reference; plus one TrainStep(mode="single_group") step equal to the drop-in path."""
    import types
    from cdcmdr_amd.model.hinet import HiNet
    d = load("g14_hinet")
    cfg = types.SimpleNamespace(use_atten=False, use_dcn=False)
    model = HiNet(FD, 4, n_tower=3, sei_dims=(16, 8), tower_dims=(8, 4), domain_idx=2, device=cuda, dropout=0.0, config=cfg).to(cuda).set_precision("f32")
    model.load_state_dict(sd_of(d))
    names = set(model.state_dict().keys())
    x = torch.from_numpy(d["x"]).to(cuda)
    y = torch.from_numpy(d["y"]).to(cuda)
    g = torch.from_numpy(d["group"]).to(cuda)
    model.train()
    pred, tt = model(x, g, targets=y)
    assert tt is y
    bce = torch.nn.BCELoss()(pred, tt.reshape(-1).float())
    reg = model.get_regularization_loss(device=cuda)
    model.zero_grad()
    (bce + reg).backward()
    assert_close(pred, d["train_pred"], RTOL, ATOL, "train_pred")
    assert_close(bce, d["bce"], RTOL, ATOL, "bce")
    assert_close(reg.reshape(-1), d["reg"].reshape(-1), 1e-5, 1e-7, "reg")
    check_grads(model, d, names)
    sd = model.state_dict()
    for k in d.files:
        if k.startswith("sd_after/"):
            assert_close(sd[k[9:]], d[k], RTOL, ATOL, k)
    model.eval()
    with torch.no_grad():
        assert_close(model(x, g)[0], d["eval_pred"], RTOL, ATOL, "eval_pred")
    # fast path: same loss on the same parameters
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    model.load_state_dict(sd_of(d))
    ts = TrainStep(model, FusedAdam(model, table_mode="dense"), x.shape[0], mode="single_group")
    got, _ = ts.step(x, y, g)
    assert_close(got, d["bce"].reshape(1), 1e-4, 1e-6, "fast-path bce")


def test_adl_matches_reference_golden(cuda):
    """G17: ADL — routing by the moving cluster centres, rows partitioned by their tower, per-tower BatchNorm statistics,
    fused output layers; training forward (predictions in tower order + permuted targets), every gradient, the centres after
    each call, BatchNorm statistics, `is_training=False` forward in batch order."""
    from cdcmdr_amd.model.adl import ADL
    d = load("g17_adl")
    model = ADL(FD, 4, n_tower=3, tower_dims=(16, 8), domain_idx=2, dropout=0.0, device=cuda, config=_atten_cfg_off()).to(cuda).set_precision("f32")
    model.load_state_dict(sd_of(d))
    assert "cluster_centers" not in model.state_dict()                     # a plain attribute in the reference
    model.cluster_centers = torch.from_numpy(d["centers0"]).to(cuda)
    names = set(model.state_dict().keys())
    x = torch.from_numpy(d["x"]).to(cuda)
    y = torch.from_numpy(d["y"]).to(cuda)
    model.train()
    pred, tt = model(x, None, targets=y, is_training=True)
    assert torch.equal(tt.cpu(), torch.from_numpy(d["train_targets"]))     # same routing, stable partition
    assert_close(model.cluster_centers, d["centers1"], 1e-5, 1e-6, "centres after the training forward")
    bce = torch.nn.BCELoss()(pred.squeeze(1), tt.reshape(-1).float())
    reg = model.get_regularization_loss(device=cuda)
    model.zero_grad()
    (bce + reg).backward()
    assert_close(pred, d["train_pred"], RTOL, ATOL, "train_pred")
    assert_close(bce, d["bce"], RTOL, ATOL, "bce")
    assert_close(reg.reshape(-1), d["reg"].reshape(-1), 1e-5, 1e-7, "reg")
    check_grads(model, d, names)
    sd = model.state_dict()
    for k in d.files:
        if k.startswith("sd_after/"):
            assert_close(sd[k[9:]], d[k], RTOL, ATOL, k)
    model.eval()
    with torch.no_grad():
        ev = model(x, is_training=False)
    assert_close(ev, d["eval_pred"], RTOL, ATOL, "eval_pred")
    assert_close(model.cluster_centers, d["centers2"], 1e-5, 1e-6, "centres after the eval forward")
