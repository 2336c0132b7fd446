"""CDC on the HIP path: the two extra step flavours its loop needs — the warm-up step on the tower MEAN
(cdc.py:100-102) and training with the module in EVAL mode (the reference's loop leaves the model in eval() after its first
evaluation pass, run.py:550) — against the oracle driven by torch autograd + torch.optim.Adam, and the whole loop
(CDCTrainer: warm-up, matrix update with snapshot/restore, regrouping, per-domain steps) as a functional run."""
import numpy as np
import pytest
import torch

from helpers import O, assert_close, is_pre_bn_bias, make_ids

pytestmark = pytest.mark.gpu
FD = [7, 400, 3, 50, 11, 29]


def _oracle_step(sd, X, y, group, mode, training, l2):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    s2 = dict(sd)
    s2.update(leaves)
    stats = {}
    y_cat = O.mmoe_forward(s2, X, FD, 3, training=training, stats_out=stats)
    p = y_cat.mean(dim=1) if mode == "mean" else y_cat.gather(1, group.reshape(-1, 1)).squeeze(1)
    loss = O.bce_mean(p, y.float()) + O.reg_loss(s2, l2)
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    loss.sum().backward()
    opt.step()
    out = {k: v.detach() for k, v in s2.items()}
    out.update(stats)
    return out, float(O.bce_mean(p, y.float()).detach())


@pytest.mark.parametrize("mode,train_mode", [("mean", True), ("multi", False), ("mean", False)])
def test_cdc_step_flavours_match_the_oracle(cuda, mode, train_mode):
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    torch.manual_seed(4)
    model = MMoE(FD, 4, 3, 4, (32, 16), (8,), dropout=0.3).to(cuda).set_precision("f32")
    for m in model.modules():
        if getattr(m, "running_mean", None) is not None:
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    if train_mode:
        model.dropout_p = 0.0                       # the oracle has no counter-based dropout stream; eval mode has none anyway
        for m in model.modules():
            if hasattr(m, "dropout_p"):
                m.dropout_p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(2)
    B = 48
    X = make_ids(rng, B, FD)
    y = rng.integers(0, 2, size=B).astype(np.int16)
    g = rng.integers(0, 3, size=B).astype(np.int64)
    model.eval()                                    # TrainStep must not depend on (or disturb) the module flag
    opt = FusedAdam(model, table_mode="dense")
    ts = TrainStep(model, opt, B, mode=mode, train_mode=train_mode)
    assert not model.training
    bce, _ = ts.step(torch.from_numpy(X).to(cuda), torch.from_numpy(y).to(cuda), None if mode == "mean" else torch.from_numpy(g).to(cuda))
    l2 = {k: 1e-5 for k in O.reg_names(list(sd), "mmoe")}
    want, want_bce = _oracle_step(sd, X, torch.from_numpy(y), torch.from_numpy(g), mode, train_mode, l2)
    assert abs(float(bce.item()) - want_bce) < 1e-5
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    names = set(sd)
    for k in names:
        if "num_batches" in k:
            continue
        if train_mode and is_pre_bn_bias(k, names):
            continue                                # noise gradients in train mode (tests/test_oracle_golden.py); real ones in eval mode
        if not train_mode and "running_" in k:
            assert torch.equal(got[k], sd[k]), f"{k}: eval-mode training must not touch the running statistics"
            continue
        assert_close(got[k], want[k], 2e-4, 2e-6, f"{mode}/train_mode={train_mode}: {k}")


def test_cdc_training_loop_runs_end_to_end(cuda, tmp_path, monkeypatch):
    import types
    from cdcmdr_amd.cdc_trainer import CDCTrainer
    from cdcmdr_amd.data import make_domain_loaders
    from cdcmdr_amd.model.cdc import CDC
    from cdcmdr_amd.optim import FusedAdam
    monkeypatch.chdir(tmp_path)
    n_domain, n_cluster, domain_idx, bs = 6, 2, 4, 64
    fd = [7, 300, 3, 50, n_domain, 29]
    rng = np.random.default_rng(0)
    n = 1500
    X = torch.from_numpy(make_ids(rng, n, fd))
    y = torch.from_numpy(rng.integers(0, 2, size=(n, 1)).astype(np.int16))
    np.random.seed(1)
    torch.manual_seed(1)
    loaders, seq, w = make_domain_loaders(X, y, bs, cuda, domain_idx, n_domain)
    cfg = types.SimpleNamespace(mmoe_n_expert=3, dataset_name="t", p_weight=0.5, p_weight_method="linear_decay", p_weight_exp_decay=0.9,
                                old_matrix_weight=0.3, affinity_func="minus", use_atten=False)
    cdc = CDC(fd, 4, n_cluster, n_domain, "mmoe", (16, 8), (8,), domain_idx, domain_cnt_weight=w, n_causal_mask=3, use_metric="loss",
              device=cuda, dropout=0.2, config=cfg).to(cuda).set_precision("f32")
    opt = FusedAdam(cdc.base_model_instance, table_mode="lazy")
    logged = []
    tr = CDCTrainer(cdc, opt, bs, loaders, n_domain, w, seq, warmup_step=1, update_matrix_step=1, update_interval=12, log=logged.append)
    tr.warmup_step, tr.update_matrix_step, tr.update_interval = 6, 2, 12          # tiny counts for the test
    table0 = cdc.base_model_instance.embedding.embedding_dict.weight.detach().clone()
    steps = tr.train_epoch(0)
    assert steps == len(seq)
    assert cdc.call_update_group == 1 + len(seq) // 12
    assert sorted(set(int(v) for v in cdc.domain2group_list)) == [0, 1] and len(cdc.domain2group_list) == n_domain
    assert all(len(s) >= 2 for s in cdc.s_group2domain_list)
    for m in (cdc.matrix_A, cdc.matrix_B, cdc.matrix_causal):
        assert bool(torch.isfinite(torch.as_tensor(m)).all())
    assert not tr.training and not cdc.training           # the mirrored quirk: eval mode since the first evaluation pass
    opt.flush_table()
    table1 = cdc.base_model_instance.embedding.embedding_dict.weight.detach()
    assert bool(torch.isfinite(table1).all()) and float((table1 - table0).abs().max()) > 1e-3
    for p in cdc.parameters():
        assert bool(torch.isfinite(p).all())
    tr.train_epoch(1)                                        # no warm-up, no forced update at i == 0
    assert tr.training is False or tr.training is True


def test_update_matrix_keeps_dense_and_lazy_table_moments_bit_equal(cuda, tmp_path, monkeypatch):
    """update_matrix() probes (train k steps, evaluate, restore the WEIGHTS): the reference's dense Adam advances the moments
    of every table row during the probes and only the weights come back (run.py:528-594, cdc.py:343-354).  The lazy table
    must land on the same moments: exact replay (fast_replay=False) == the dense table mode, bit for bit."""
    import types
    from cdcmdr_amd.cdc_trainer import CDCTrainer
    from cdcmdr_amd.data import make_domain_loaders
    from cdcmdr_amd.model.cdc import CDC
    from cdcmdr_amd.optim import FusedAdam
    monkeypatch.chdir(tmp_path)
    n_domain, n_cluster, domain_idx, bs = 6, 2, 4, 64
    fd = [7, 300, 3, 50, n_domain, 29]
    res = {}
    for table_mode in ("dense", "lazy"):
        rng = np.random.default_rng(0)
        n = 1500
        X = torch.from_numpy(make_ids(rng, n, fd))
        y = torch.from_numpy(rng.integers(0, 2, size=(n, 1)).astype(np.int16))
        np.random.seed(1)
        torch.manual_seed(1)
        loaders, seq, w = make_domain_loaders(X, y, bs, cuda, domain_idx, n_domain)
        cfg = types.SimpleNamespace(mmoe_n_expert=3, dataset_name="t", p_weight=0.5, p_weight_method="linear_decay", p_weight_exp_decay=0.9,
                                    old_matrix_weight=0.3, affinity_func="minus", use_atten=False)
        cdc = CDC(fd, 4, n_cluster, n_domain, "mmoe", (16, 8), (8,), domain_idx, domain_cnt_weight=w, n_causal_mask=2, use_metric="loss",
                  device=cuda, dropout=0.0, config=cfg).to(cuda).set_precision("f32")
        opt = FusedAdam(cdc.base_model_instance, table_mode=table_mode, fast_replay=False, flush_every=4)
        tr = CDCTrainer(cdc, opt, bs, loaders, n_domain, w, seq, warmup_step=1, update_matrix_step=1, update_interval=0)
        tr.warmup_step, tr.update_matrix_step = 3, 2
        for _ in range(3):                                        # a few ordinary steps first, so the moments are not all zero
            Xb, yb = tr.get_domain_data(0)
            tr._step(Xb, yb, "split", domain_i=0)
        tr.update_matrix()
        opt.flush_table()
        res[table_mode] = (opt.table_m.cpu().clone(), opt.table_v.cpu().clone(),
                           cdc.base_model_instance.embedding.embedding_dict.weight.detach().cpu().clone(), int(opt.step_dev.item()))
    assert res["dense"][3] == res["lazy"][3] > 3
    assert torch.equal(res["dense"][2], res["lazy"][2]), "table weights differ"
    assert torch.equal(res["dense"][0], res["lazy"][0]), "exp_avg of the table differs after update_matrix()"
    assert torch.equal(res["dense"][1], res["lazy"][1]), "exp_avg_sq of the table differs after update_matrix()"
