"""BASELINE configurations C4 and C5 as CONFIGURATIONS (missing #2 / #4):

  C4  CDC(base='ple'): 30 domains -> 4 clusters, emb_dim 32, nested expert dims (model/cdc.py:32-42), trained in split mode
      (run.py:635-640) — three steps against a golden captured from the reference itself (tools/make_golden.py: g5_cdc_ple_adam),
      on the dense and on the lazy table; and the table at its FULL size (26 x 10 M x 32 fp32 = 33 GB + 67 GB of moments) through
      size-independent properties on one GPU.
  C5  STAR with 30 towers, bf16 contractions, GROUPED mode (rows partitioned by domain, ragged groups incl. an empty one and a
      one-row one: model/star.py:84-114, run.py:477-480) — a training step against the oracle's bf16 restatement.
"""
import os
import types

import numpy as np
import pytest
import torch

from helpers import O, assert_close, is_pre_bn_bias, make_ids, sd_cpu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cdc_config():
    return types.SimpleNamespace(mmoe_n_expert=4, ple_n_expert_specific=2, ple_n_expert_shared=2, gate_hidden_dim=8,
                                 dataset_name="golden", p_weight=0.5, p_weight_method="none", old_matrix_weight=0.0,
                                 affinity_func="minus", use_atten=False, n_cross_layers=3)


def _assert_close_after_adam(got, want, n_steps, what, lr=1e-3):
    """Parameters after n Adam steps: rtol 5e-5 / atol 5e-6 like the other trajectory tests — except that an element whose
    gradient is at rounding level (|g| ~ eps = 1e-8: here the experts of the clusters a single-domain batch does not train, which
    only see the batch through the shared gate) is moved by lr * g / (|g| + eps), anywhere within +-lr per step depending on
    the last bits of g.  At most 0.5 % of a tensor's elements may be such, and they stay within n * lr of the reference."""
    a = torch.as_tensor(got).detach().cpu().double()
    b = torch.as_tensor(want).detach().cpu().double()
    assert a.shape == b.shape, what
    err = (a - b).abs()
    bad = err > 5e-6 + 5e-5 * b.abs()
    assert int(bad.sum()) <= max(1, int(0.005 * a.numel())), f"{what}: {int(bad.sum())}/{a.numel()} elements off"
    assert float(err.max()) <= n_steps * lr * 1.01 + 5e-6, f"{what}: an element moved {float(err.max()):.3e} away from the reference"


@pytest.mark.parametrize("table_mode", ["dense", "lazy"])
def test_c4_cdc_ple_split_mode_training_matches_the_reference(cuda, table_mode, tmp_path, monkeypatch):
    from cdcmdr_amd.model.cdc import CDC
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    monkeypatch.chdir(tmp_path)
    d = np.load(os.path.join(GOLD, "g5_cdc_ple_adam.npz"))
    fd, d2g, dom_idx = d["field_dims"].tolist(), d["domain2group"], int(d["domain_idx"])
    n_dom, n_clu, D, B = len(d2g), int(d2g.max()) + 1, 32, d["x0"].shape[0]
    assert (n_dom, n_clu) == (30, 4)
    cdc = CDC(fd, D, n_clu, n_dom, "ple", ((32, 16), (8,)), (8, 4), dom_idx, domain_cnt_weight=np.full(n_dom, 1.0 / n_dom),
              n_causal_mask=4, device=cuda, dropout=0.0, config=_cdc_config()).to(cuda).set_precision("f32")
    cdc.load_state_dict({k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd0/")})
    cdc.domain2group = torch.from_numpy(d2g).to(cuda)
    cdc.domain2group_list = d2g.tolist()
    base = cdc.base_model_instance
    opt = FusedAdam(base, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, table_mode=table_mode)
    ts = TrainStep(base, opt, B, mode="multi")
    names = list(cdc.state_dict().keys())
    params = dict(cdc.named_parameters())
    for s in range(3):
        X = torch.from_numpy(d[f"x{s}"]).to(cuda)
        y = torch.from_numpy(d[f"y{s}"]).to(cuda)
        dom = int(d[f"domain{s}"])
        # mode='split': with domain_i every row trains its domain's cluster tower (cdc.py:109-111), without it the tower of the
        # cluster of the row's own domain (cdc.py:106-107) — both are "group = domain2group[domain of the row]"
        group = cdc.groups_of(X)
        if dom >= 0:
            assert bool((group == int(d2g[dom])).all())
        if table_mode == "lazy":
            ts.refresh_table_reg()
        bce, reg = ts.step(X, y, group)
        assert_close(bce, d[f"bce{s}"].reshape(1), 1e-4, 1e-6, f"bce{s}")
        assert_close(reg.reshape(1), d[f"reg{s}"].reshape(1), 1e-5, 1e-7, f"reg{s}")
        opt.flush_table()
        sd = cdc.state_dict()
        for k in names:
            gk = f"sd{s + 1}/{k}"
            if is_pre_bn_bias(k.replace("base_model_instance.", ""), {n.replace("base_model_instance.", "") for n in names}):
                params[k].data.copy_(torch.from_numpy(d[gk]))           # noise gradient whose sign Adam turns into +-lr (test_gpu_train.py)
                continue
            _assert_close_after_adam(sd[k], d[gk], s + 1, gk)
    t = "base_model_instance.embedding.embedding_dict.weight"
    assert_close(opt.table_m, d[f"m3/{t}"], 1e-3, 1e-7, "table exp_avg after 3 steps")
    # F3 on the C4 row shape: rows 701..899 of field 1 were never looked up and moved ~lr per step like the reference's
    lo, hi = fd[0] + 701, fd[0] + 900
    w0, w3 = d[f"sd0/{t}"][lo:hi], cdc.state_dict()[t][lo:hi].cpu().numpy()
    moved = np.abs(w3 - w0)
    big = np.abs(w0) > 0.05                                           # (an element within 3 lr of zero turns around on the way)
    assert moved[big].min() > 2.5e-3 and moved.max() < 3.5e-3
    assert_close(w3, d[f"sd3/{t}"][lo:hi], 1e-6, 1e-7, "untouched rows after 3 steps")


def test_c4_full_size_table_properties(cuda):
    """26 fields x vocab 10 M x emb_dim 32: a 260 M-row table (33 GB) + Adam moments (67 GB) + the rows' step stamps on ONE GPU.
    Size-independent properties: the gather is an exact row copy at the far end of the table (row offsets beyond 2^31 BYTES and a
    flat element index beyond 2^32), two identical runs give identical bits (losses, dense parameters, sampled rows and moments),
    rows no batch looked up follow the L2-only recurrence (they moved ~lr per step), and an out-of-range id is flagged."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    from cdcmdr_amd.model.cdc import CDC
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    F_, V, D, B, n_dom, n_clu = 26, 10_000_000, 32, 1024, 30, 4
    free, _ = torch.cuda.mem_get_info()
    need = F_ * V * D * 4 * 3 + F_ * V * 4 + (8 << 30)
    if free < need:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of device memory for the full-size C4 table, {free / 2**30:.0f} GiB free")
    fd = [V] * F_
    dom_idx = 10
    rng = np.random.default_rng(5)
    d2g = torch.tensor([(3 * k + 1) % n_clu for k in range(n_dom)], dtype=torch.int64, device=cuda)

    def batches(n):
        X = rng.integers(0, V, size=(n, B, F_), dtype=np.int64).astype(np.int32)
        X[:, :, dom_idx] = rng.integers(0, n_dom, size=(n, B))
        X[:, :, F_ - 1] = V - 1 - rng.integers(0, 1000, size=(n, B))            # the far end of the table
        y = rng.integers(0, 2, size=(n, B)).astype(np.int16)
        return X, y
    Xs, ys = batches(3)

    def run():
        torch.manual_seed(2000)
        with torch.device(cuda):
            cdc = CDC(fd, D, n_clu, n_dom, "ple", ((256, 128), (64,)), (64, 32), dom_idx, n_causal_mask=4, device=cuda, dropout=0.2,
                      config=_cdc_config())
        cdc.set_precision("bf16")
        cdc.domain2group = d2g
        base = cdc.base_model_instance
        table = base.embedding.embedding_dict.weight
        w0 = table.detach()[:4096].clone()
        opt = FusedAdam(base, table_mode="lazy")
        ts = TrainStep(base, opt, B, mode="multi", use_graph=True)
        losses = []
        for s in range(3):
            X = torch.from_numpy(Xs[s]).to(cuda)
            bce, _ = ts.step(X, torch.from_numpy(ys[s]).to(cuda), cdc.groups_of(X))
            losses.append(bce.clone())
        ts.check_ids()
        opt.flush_table()
        rows = torch.from_numpy((Xs.astype(np.int64) + np.arange(F_, dtype=np.int64) * V).reshape(-1)[::7].copy()).to(cuda)
        out = {"losses": torch.stack(losses).cpu(), "rows": table.detach()[rows].cpu(), "m": opt.table_m[rows].cpu(),
               "v": opt.table_v[rows].cpu(), "head": table.detach()[:4096].cpu(), "w0": w0.cpu(),
               "dense": {k: v.detach().cpu() for k, v in base.state_dict().items() if "embedding_dict" not in k}}
        if run.first:
            run.first = False
            # the gather at the far end of the table: an exact copy of rows whose byte offset is beyond 2^35
            lib = L.load()
            X = torch.from_numpy(Xs[0]).to(cuda)
            got = torch.empty((B, F_ * D), dtype=torch.float32, device=cuda)
            offs = base.embedding.offsets_device(cuda)
            L.check(lib.cdc_embed_gather_fwd(X.data_ptr(), offs.data_ptr(), table.data_ptr(), got.data_ptr(), None, None, B, F_, D,
                                             table.shape[0], C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gather")
            flat = (X.long() + torch.arange(F_, device=cuda) * V).reshape(-1)
            assert int(flat.max()) * D > 2 ** 32
            assert torch.equal(got.view(-1, D), table.detach()[flat])
            # an id outside its field is flagged like the reference's IndexError
            bad = X.clone()
            bad[3, F_ - 1] = V                                      # (one past the LAST field: past the end of the table, like nn.Embedding's check)
            ts.step(bad, torch.from_numpy(ys[0]).to(cuda), cdc.groups_of(X))
            with pytest.raises(IndexError):
                ts.check_ids()
        del ts, opt, cdc, base, table
        torch.cuda.empty_cache()
        return out
    run.first = True
    a = run()
    b = run()
    assert torch.equal(a["losses"], b["losses"]) and bool(torch.isfinite(a["losses"]).all())
    for k in ("rows", "m", "v", "head"):
        assert torch.equal(a[k], b[k]), f"table {k} differs between two identical runs"
    for k in a["dense"]:
        assert torch.equal(a["dense"][k], b["dense"][k]), f"{k} differs between two identical runs"
    # rows 0..4095 of field 0: with 3 x 1024 uniform draws out of 10 M nearly all of them were never looked up -> ~lr per step (F3)
    moved = (a["head"] - a["w0"]).abs()
    frac = float(((moved > 2.5e-3) & (moved < 3.5e-3)).float().mean())
    assert frac > 0.99, f"only {frac:.3f} of the untouched elements moved by ~3 lr"


def test_whole_table_flush_beyond_2_31_work_items(cuda):
    """The replay kernel counts a launch's work items (rows x 16-byte chunks) in 32 bits.  134 218 728 rows x emb_dim 64 =
    2^31 + 16 000 items: the whole-table flush (flush_table / state_dict / evaluation) has to go out as consecutive row windows.
    Every element starts from the same state, so every row must end at the value a small table reaches through the same
    entry point; a skipped window shows as untouched rows, a wrapped index as a fault or as stale stamps."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    from cdcmdr_amd.model.dcn import DCN
    from cdcmdr_amd.optim import FusedAdam
    R, D, Rs, target = (1 << 27) + 1000, 64, 4096, 5
    free, _ = torch.cuda.mem_get_info()
    need = 3 * R * D * 4 + R * 4 + (6 << 30)
    if free < need:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of device memory, {free / 2**30:.0f} GiB free")
    assert R * (D // 4) > 2 ** 31
    with torch.device(cuda):
        small_model = DCN([16] * 3, 4, 2, (8,), dropout=0.0)
    opt = FusedAdam(small_model, table_mode="lazy")                     # (its hyper-parameter block and step-scalar tables)
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    step = torch.full((1,), target, dtype=torch.int32, device=cuda)
    res = {}
    for rows in (Rs, R):
        w = torch.full((rows, D), 0.05, dtype=torch.float32, device=cuda)
        m = torch.full((rows, D), 1e-7, dtype=torch.float32, device=cuda)
        v = torch.full((rows, D), 1e-12, dtype=torch.float32, device=cuda)
        last = torch.zeros(rows, dtype=torch.int32, device=cuda)
        L.check(lib.cdc_embed_lazy_flush(w.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), rows, D, opt._hp(), step.data_ptr(), 0, 0,
                                         0, 0, st), "flush")
        torch.cuda.synchronize()
        if rows == Rs:
            res = {"w": w[0].clone(), "m": m[0].clone(), "v": v[0].clone()}
            assert float((w[0] - 0.05).abs().max()) > 1e-3               # five L2-only Adam steps moved it by ~5 lr
            assert torch.equal(w, res["w"].expand_as(w))
            continue
        assert int(last.min()) == target and int(last.max()) == target
        for name, t in (("w", w), ("m", m), ("v", v)):
            want = res[name]
            for lo in range(0, rows, 1 << 23):                           # 8 M rows per comparison
                blk = t[lo:lo + (1 << 23)]
                assert bool((blk == want).all()), f"{name}: rows in [{lo}, {lo + blk.shape[0]}) differ from the small table's result"
        del w, m, v, last
    # a SLICE launch (rows chosen on the device) of that size is refused, not wrapped
    one = C.c_void_p(256)
    rc = lib.cdc_embed_lazy_flush(one, one, one, one, C.c_int64(1 << 40), 4, opt._hp(), step.data_ptr(), 0, 2, 0, 0, st)
    assert rc == -2 and b"exceeds" in lib.cdc_last_error()


def test_c5_star30_bf16_grouped_training_step(cuda):
    """STAR, 30 towers, bf16 contractions, GROUPED mode: one TrainStep(mode='star') on a batch whose rows are partitioned by domain
    into ragged groups — domain 7 is absent (empty group: the tower is skipped and BatchNorm keeps its statistics, star.py:94),
    domain 11 has exactly one row (BatchNorm skipped for that group, star.py:134-135) — against the oracle's bf16 restatement
    (operands of every contraction rounded where the kernels round them) + torch.optim.Adam with the reference's settings."""
    from cdcmdr_amd.model.star import STAR
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    n_tower, B = 30, 1024
    fd = [50, 3000, 11, 700, n_tower, 29]
    dom_idx = 4
    torch.manual_seed(8)
    model = STAR(fd, 16, n_tower, (64, 32, 16), domain_idx=dom_idx, dropout=0.0).to(cuda).set_precision("bf16")
    sd = sd_cpu(model)
    rng = np.random.default_rng(9)
    X = make_ids(rng, B, fd)
    dom = X[:, dom_idx]
    dom[dom == 7] = 8
    dom[dom == 11] = 12
    dom[5] = 11
    y = rng.integers(0, 2, size=B).astype(np.int16)
    g = X[:, dom_idx].astype(np.int64)
    counts = np.bincount(g, minlength=n_tower)
    assert counts[7] == 0 and counts[11] == 1 and counts.max() > 8
    opt = FusedAdam(model, table_mode="lazy")
    ts = TrainStep(model, opt, B, mode="star")
    bce, _ = ts.step(torch.from_numpy(X).to(cuda), torch.from_numpy(y).to(cuda), torch.from_numpy(g).to(cuda))
    opt.flush_table()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    s2 = dict(sd)
    s2.update(leaves)
    l2 = {k: 1e-5 for k in O.reg_names(list(sd), "star")}
    ref_opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    stats = {}
    O.MATMUL_BF16 = "exact"
    try:
        p, t = O.star_forward(s2, X, fd, n_tower, x_group=g, targets=torch.from_numpy(y).float(), training=True, stats_out=stats)
        want_bce = O.bce_mean(p.squeeze(1), t)
        keys = list(leaves)
        g_bce = dict(zip(keys, torch.autograd.grad(want_bce, [leaves[k] for k in keys], retain_graph=True, allow_unused=True)))
        (want_bce + O.reg_loss(s2, l2).sum()).backward()
    finally:
        O.MATMUL_BF16 = False
    assert abs(float(bce.item()) - float(want_bce.detach())) < 2e-4
    # the dense-parameter gradients of the BCE term (the L2 term is folded into the optimiser kernels) against the restatement's,
    # under the bounds every bf16 path is held to (tests/helpers.py)
    from helpers import compare_param_grads
    pg = ts.plan.param_grads
    have = {k: types.SimpleNamespace(grad=pg[id(p_)]) for k, p_ in model.named_parameters() if id(p_) in pg}
    compare_param_grads(have, {k: v for k, v in g_bce.items() if k in have}, 5e-3, 2e-3, bf16=True, all_names=list(sd))
    grads = {k: (leaf.grad.clone() if leaf.grad is not None else None) for k, leaf in leaves.items()}
    # one Adam step: the sign of a gradient decides the first move (+-lr), so compare parameters through the update they imply:
    # a parameter whose gradient is clearly non-zero on the oracle side must have moved the same way
    ref_opt.step()
    got = sd_cpu(model)
    names = set(sd)
    checked = agree = 0
    for k, leaf in leaves.items():
        if grads[k] is None or "num_batches" in k or is_pre_bn_bias(k, names):
            continue
        if k == "shared_bn_bias" or (k.startswith("domain_norm.") and k.endswith(".bias")):
            continue
        gsc = float(grads[k].abs().max())
        if gsc == 0.0:
            continue
        clear = grads[k].abs() > 0.05 * gsc                               # gradient elements well away from zero
        if "embedding_dict" in k:
            clear &= (grads[k].abs() > 2e-6)
        d_got = (got[k] - sd[k])[clear]
        d_want = (leaf.detach() - sd[k])[clear]
        checked += int(clear.sum())
        agree += int((torch.sign(d_got) == torch.sign(d_want)).sum())
    assert checked > 1000 and agree / checked > 0.995, f"{agree}/{checked} clearly-signed elements moved the same way"
    # the absent domain's tower: no BCE gradient at all (its parameters still move by the L2 term, like the reference's)
    for k, v in g_bce.items():
        if k.startswith("domain_dnns.7.") or k.startswith("domain_dnn_linears.7."):
            assert v is None or float(v.abs().max()) == 0.0, k
            if k in have:
                assert float(have[k].grad.abs().max()) == 0.0, f"{k}: the absent domain's tower received a gradient"
    # BatchNorm statistics: the empty group's stay untouched, the others match the restatement
    for k, v in stats.items():
        assert_close(got[k], v, 5e-3, 2e-3, f"stat {k}")
    for k in names:
        if k.startswith("domain_dnns.7.") and "running_" in k:
            assert torch.equal(got[k], sd[k]), f"{k}: the absent domain's statistics moved"
