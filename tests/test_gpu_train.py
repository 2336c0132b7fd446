"""The training step on the HIP path (TrainStep + FusedAdam) against the reference's own three-step golden
trajectories (G3) and the CPU oracle: parameters, Adam moments, losses, BatchNorm statistics, the never-touched
table rows, dense vs lazy table modes, eager vs hipGraph replay."""
import os

import numpy as np
import pytest
import torch

from helpers import O, assert_close, is_pre_bn_bias, make_ids

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FD_SPARSE = [7, 400, 3, 50, 11, 29]


def _sd_of(d, prefix):
    p = prefix + "/"
    return {k[len(p):]: torch.from_numpy(np.asarray(d[k])) for k in d.files if k.startswith(p)}


def _make(kind, cuda, precision="f32"):
    if kind == "ple":
        from cdcmdr_amd.model.ple import PLE
        m = PLE(FD_SPARSE, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0)
    else:
        from cdcmdr_amd.model.dcn import DCN
        m = DCN(FD_SPARSE, 4, 3, (32, 16, 8), dropout=0.0)
    return m.to(cuda).set_precision(precision)


@pytest.mark.parametrize("kind,gold", [("ple", "g3_ple3_adam"), ("dcn", "g3_dcn_adam")])
@pytest.mark.parametrize("table_mode", ["dense", "lazy"])
def test_three_steps_match_reference_golden(cuda, kind, gold, table_mode):
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    d = np.load(os.path.join(GOLD, gold + ".npz"))
    model = _make(kind, cuda)
    model.load_state_dict(_sd_of(d, "sd0"))
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, table_mode=table_mode)
    ts = TrainStep(model, opt, 32, mode="multi" if kind == "ple" else "single")
    names = list(model.state_dict().keys())
    params = dict(model.named_parameters())
    for s in range(3):
        X = torch.from_numpy(d[f"x{s}"]).to(cuda)
        y = torch.from_numpy(d[f"y{s}"]).to(cuda)
        g = torch.from_numpy(d[f"group{s}"]).to(cuda) if f"group{s}" in d.files else None
        if table_mode == "lazy":
            ts.refresh_table_reg()                # the table's l2 * sum(w^2) for the weights this step's forward sees (flush + sum)
        bce, reg = ts.step(X, y, g)
        assert_close(bce, d[f"bce{s}"].reshape(1), 1e-4, 1e-6, f"bce{s}")
        assert_close(reg.reshape(1), d[f"reg{s}"].reshape(1), 1e-5, 1e-7, f"reg{s}")      # run.py:489's term, both table modes
        opt.flush_table()
        sd = model.state_dict()
        for k in names:
            gk = f"sd{s + 1}/{k}"
            if is_pre_bn_bias(k, set(names)):
                # rounding-noise gradient whose SIGN Adam turns into a +-lr move (see tests/test_oracle_golden.py):
                # adopt the reference's value and moments so the statistics that contain it stay comparable
                params[k].data.copy_(torch.from_numpy(d[gk]))
                st = opt.state[id(params[k])]
                st[0].copy_(torch.from_numpy(d[f"m{s + 1}/{k}"]))
                st[1].copy_(torch.from_numpy(d[f"v{s + 1}/{k}"]))
                continue
            assert_close(sd[k], d[gk], 5e-5, 5e-6, gk)
        for k, p in params.items():
            if is_pre_bn_bias(k, set(names)) or f"m{s + 1}/{k}" not in d.files:
                continue
            m_, v_ = (opt.table_m, opt.table_v) if p is opt.table else opt.state[id(p)]
            assert_close(m_, d[f"m{s + 1}/{k}"], 1e-3, 1e-7, f"m{s + 1}/{k}")
            assert_close(v_, d[f"v{s + 1}/{k}"], 2e-3, 1e-10, f"v{s + 1}/{k}")
    # F3: never-looked-up rows moved by ~lr per step, exactly like the reference's dense Adam + whole-table L2
    w0 = d["sd0/embedding.embedding_dict.weight"][7 + 301:7 + 400]
    w3 = model.embedding.embedding_dict.weight.detach().cpu().numpy()[7 + 301:7 + 400]
    moved = np.abs(w3 - w0)
    assert moved.min() > 2.5e-3 and moved.max() < 3.5e-3
    assert_close(w3, d["sd3/embedding.embedding_dict.weight"][7 + 301:7 + 400], 1e-6, 1e-7, "untouched rows after 3 steps")


def _train_table(cuda, n_steps=12, **opt_kw):
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = [50, 3000, 7, 900]
    torch.manual_seed(11)
    model = PLE(fd, 8, 3, 1, 1, ((16,), (8,)), (8,), dropout=0.0).to(cuda).set_precision("f32")
    opt = FusedAdam(model, **opt_kw)
    ts = TrainStep(model, opt, 128)
    r = np.random.default_rng(5)
    for _ in range(n_steps):
        X = torch.from_numpy(make_ids(r, 128, fd)).to(cuda)
        y = torch.from_numpy(r.integers(0, 2, size=128).astype(np.int16)).to(cuda)
        g = torch.from_numpy(r.integers(0, 3, size=128).astype(np.int64)).to(cuda)
        ts.step(X, y, g)
    opt.flush_table()
    return model.embedding.embedding_dict.weight.detach().cpu().clone(), opt.table_m.cpu().clone(), opt.table_v.cpu().clone()


def test_lazy_fast_replay_and_periodic_flush_stay_on_the_exact_trajectory(cuda):
    """Default lazy mode (hardware rcp/sqrt in the replay, whole-table catch-up every few steps) vs the exact dense mode
    after 40 steps: the table differs by at most a few 1e-7 (fp32 ulp of O(1) weights = 1.2e-7), Adam moments alike."""
    dense = _train_table(cuda, n_steps=40, table_mode="dense")
    for kw in (dict(fast_replay=True, flush_every=0), dict(fast_replay=True, flush_every=8), dict(fast_replay=False, flush_every=8)):
        lazy = _train_table(cuda, n_steps=40, table_mode="lazy", **kw)
        dw = float((lazy[0] - dense[0]).abs().max())
        assert dw <= (0.0 if not kw["fast_replay"] else 1e-6), f"{kw}: table differs from the exact trajectory by {dw:.2e}"
        assert_close(lazy[1], dense[1], 1e-4, 1e-9, f"m {kw}")
        assert_close(lazy[2], dense[2], 1e-4, 1e-12, f"v {kw}")


def test_fast_replay_in_many_short_segments_does_not_drift(cuda):
    """The scaled-state replay turns a row's moments into M = m * ik, V = v * ik2 at the start of a segment and back at its end.
    With the way back as `* k` (k the fp32 scale, ik = fl(1/k)) every segment left the factor ik * k = 1 +- 6e-8 on the state —
    the SAME factor on every row, every segment: 300 one-step segments moved v by 2e-5 relative to one 300-step segment, and over a
    training run that coherent drift of all moments doubled the spread of the AUC over row orders (tools/auc_spread_probe.py,
    DESIGN.md section 7).  Now the way back multiplies by k + k_lo = 1 / ik to 2^-48: 300 one-step segments and one 300-step segment
    agree to rounding noise (a random walk of half-ulp roundings: ~1e-6), with no common sign."""
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    fd = [50, 3000, 7, 900]
    K = 300
    res = {}
    for segments in ("short", "long"):
        torch.manual_seed(11)
        model = PLE(fd, 8, 3, 1, 1, ((16,), (8,)), (8,), dropout=0.0).to(cuda).set_precision("f32")
        opt = FusedAdam(model, table_mode="lazy", fast_replay=True, flush_every=0)
        assert opt.replay_tab is not None
        g = torch.Generator(device="cpu").manual_seed(3)
        R, D = opt.table.shape
        with torch.no_grad():
            opt.table.data.copy_((torch.randn(R, D, generator=g) * 0.01).to(cuda))
            opt.table_m.copy_((torch.randn(R, D, generator=g) * 1e-3).to(cuda))          # moments at the scale real gradients leave
            opt.table_v.copy_((torch.rand(R, D, generator=g) * 1e-6 + 1e-8).to(cuda))
        opt.table_last.zero_()
        opt.step_dev.zero_()
        if segments == "short":
            for _ in range(K):
                opt.step_dev += 1
                opt.flush_table()
        else:
            opt.step_dev += K
            opt.flush_table()
        torch.cuda.synchronize()
        assert int(opt.table_last.min().item()) == K
        res[segments] = (opt.table.data.double().cpu(), opt.table_v.double().cpu())
    dw = (res["short"][0] - res["long"][0]).abs()
    assert float(dw.max()) < 2e-7, f"w: one-step segments differ from one long segment by {float(dw.max()):.2e}"      # (|w| ~ 1e-2: ulp 1e-9)
    a, b = res["short"][1], res["long"][1]
    rel = (a - b) / b.abs().clamp_min(1e-30)
    # rounding noise: |rel| a few 1e-7 with both signs; the drift this guards against: +-1.8e-5 on every element of v, one sign
    print(f"v: max |rel| {float(rel.abs().max()):.2e}, mean rel {float(rel.mean()):+.2e}")
    assert float(rel.abs().max()) < 6e-6, f"v: one-step segments differ from one long segment by {float(rel.abs().max()):.2e} (relative)"
    assert abs(float(rel.mean())) < 2e-7, f"v: common relative shift {float(rel.mean()):+.2e}"


def test_dense_and_lazy_table_modes_are_bit_identical(cuda):
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    rng = np.random.default_rng(3)
    fd = [50, 3000, 7, 900]
    tables = {}
    for mode in ("dense", "lazy"):
        from cdcmdr_amd.model.ple import PLE
        torch.manual_seed(11)
        model = PLE(fd, 8, 3, 1, 1, ((16,), (8,)), (8,), dropout=0.0).to(cuda).set_precision("f32")
        opt = FusedAdam(model, table_mode=mode, fast_replay=False, flush_every=0)
        ts = TrainStep(model, opt, 128)
        r = np.random.default_rng(5)
        for _ in range(12):
            X = torch.from_numpy(make_ids(r, 128, fd)).to(cuda)
            y = torch.from_numpy(r.integers(0, 2, size=128).astype(np.int16)).to(cuda)
            g = torch.from_numpy(r.integers(0, 3, size=128).astype(np.int64)).to(cuda)
            ts.step(X, y, g)
        opt.flush_table()
        tables[mode] = (model.embedding.embedding_dict.weight.detach().cpu().clone(), opt.table_m.cpu().clone(), opt.table_v.cpu().clone())
    for a, b, what in zip(tables["dense"], tables["lazy"], ("w", "m", "v")):
        assert torch.equal(a, b), f"table {what}: lazy replay differs from the dense pass"


@pytest.mark.parametrize("table_mode", ["dense", "lazy"])
def test_graph_replay_equals_eager(cuda, table_mode):
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    from cdcmdr_amd.model.mmoe import MMoE
    fd = [20, 500, 7, 90, 4]
    res = {}
    for use_graph in (False, True):
        torch.manual_seed(4)
        model = MMoE(fd, 8, 3, 4, (32, 16), (8,), dropout=0.0).to(cuda).set_precision("f32")
        opt = FusedAdam(model, table_mode=table_mode)
        ts = TrainStep(model, opt, 64, use_graph=use_graph)
        r = np.random.default_rng(8)
        losses = []
        for _ in range(6):
            X = torch.from_numpy(make_ids(r, 64, fd)).to(cuda)
            y = torch.from_numpy(r.integers(0, 2, size=64).astype(np.int16)).to(cuda)
            g = torch.from_numpy(r.integers(0, 3, size=64).astype(np.int64)).to(cuda)
            bce, _ = ts.step(X, y, g)
            losses.append(float(bce.item()))
        opt.flush_table()
        res[use_graph] = (losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        if use_graph:
            assert ts.graph is not None
    assert res[False][0] == res[True][0]
    for k in res[False][1]:
        assert torch.equal(res[False][1][k], res[True][1][k]), k


@pytest.mark.parametrize("precision,B", [("bf16", 1024), ("f32", 512)])
def test_split_k_slabs_summed_by_the_adam_launch_train_bit_identically(cuda, monkeypatch, precision, B):
    """Single GPU: the batched grad-weight launches leave their split-K slabs unreduced and the dense Adam launch adds them in the
    order the reduce launch would (cdc_lin_bwdw_args.defer_reduce, cdc_adam_tensor.slabs).  Same sums in the same order: weights,
    moments and losses after six steps are bit-identical to the run with the reduce launches."""
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    from cdcmdr_amd.model.ple import PLE
    fd = [50, 3, 500, 7, 90, 4, 1000, 30]
    res = {}
    for defer in ("0", "1"):
        torch.manual_seed(4)
        model = PLE(fd, 16, 3, 2, 2, ((256, 128), (64,)), (64, 32), dropout=0.2).to(cuda).set_precision(precision)
        model.seed = 77
        opt = FusedAdam(model, table_mode="lazy")
        ts = TrainStep(model, opt, B, use_graph=(defer == "1"), defer_dw_reduce=(defer == "1"))
        assert bool(ts.plan.grad_slabs) == (defer == "1"), "split-K launches expected at this batch size"
        r = np.random.default_rng(8)
        losses = []
        for _ in range(6):
            X = torch.from_numpy(make_ids(r, B, fd)).to(cuda)
            y = torch.from_numpy(r.integers(0, 2, size=B).astype(np.int16)).to(cuda)
            g = torch.from_numpy(r.integers(0, 3, size=B).astype(np.int64)).to(cuda)
            bce, reg = ts.step(X, y, g)
            losses.append((float(bce.item()), float(reg.item())))
        opt.flush_table()
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        mom = {k: (v["exp_avg"].cpu(), v["exp_avg_sq"].cpu()) for k, v in opt.state_dict()["state"].items()}
        res[defer] = (losses, sd, mom)
    assert [b for b, _ in res["0"][0]] == [b for b, _ in res["1"][0]]
    # (the reg figure is a sum of per-workgroup double partials added by atomics: equal to the last bits, not bit for bit)
    assert np.allclose([r_ for _, r_ in res["0"][0]], [r_ for _, r_ in res["1"][0]], rtol=1e-12, atol=0)
    for k in res["0"][1]:
        assert torch.equal(res["0"][1][k], res["1"][1][k]), k
    for k in res["0"][2]:
        assert torch.equal(res["0"][2][k][0], res["1"][2][k][0]) and torch.equal(res["0"][2][k][1], res["1"][2][k][1]), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_row_sort_one_batch_ahead_trains_bit_identically(cuda, monkeypatch, use_graph):
    """Single GPU, lazy table: step(..., next_X=) sorts the NEXT batch's rows on the side chain of this step (trainer._sort_ahead).
    The sort reads nothing but ids, so the trajectory is bit-identical to steps that sort their own batch first — also when the
    announced batch is not the one that comes (the step then sorts its own), when no batch is announced, and across a ragged
    announcement (wrong shape: ignored)."""
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    from cdcmdr_amd.model.ple import PLE
    fd = [50, 3, 500, 7, 90, 4, 1000, 30]
    B, n = 256, 10
    r = np.random.default_rng(8)
    Xs = [torch.from_numpy(make_ids(r, B, fd)).to(cuda) for _ in range(n + 1)]
    ys = [torch.from_numpy(r.integers(0, 2, size=B).astype(np.int16)).to(cuda) for _ in range(n)]
    gs = [torch.from_numpy(r.integers(0, 3, size=B).astype(np.int64)).to(cuda) for _ in range(n)]
    decoy = torch.from_numpy(make_ids(r, B, fd)).to(cuda)
    res = {}
    for ahead in ("0", "1"):
        torch.manual_seed(4)
        model = PLE(fd, 8, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.2).to(cuda).set_precision("f32")
        model.seed = 5
        opt = FusedAdam(model, table_mode="lazy", flush_every=4)
        ts = TrainStep(model, opt, B, use_graph=use_graph, sort_ahead=(ahead == "1"))
        losses = []
        for i in range(n):
            nxt = Xs[i + 1]
            if i == 4:
                nxt = decoy                              # announced, never comes: step 5 sorts its own batch
            if i == 6:
                nxt = None                               # nothing announced
            if i == 7:
                nxt = Xs[i + 1][:B // 2]                 # wrong shape: ignored
            if i == 3:
                # another step of the same optimiser in between (a sibling for a ragged batch, as data.train_epoch issues at the end of
                # an epoch): its dense-Adam argument set must not replace (and free) the one the main step's graphs were captured with
                ts.sibling(B // 2).step(decoy[:B // 2].contiguous(), ys[0][:B // 2].contiguous(), gs[0][:B // 2].contiguous())
            bce, _ = ts.step(Xs[i], ys[i], gs[i], next_X=nxt)
            losses.append(float(bce.item()))
        # (the look-ahead rides on the side chain: TrainStep(overlap=False) or the fused catch-up + gather switch it off)
        assert bool(getattr(ts, "_ahead_ok", False)) == (ahead == "1" and ts._overlap() and not ts._fuse_gather())
        opt.flush_table()
        res[ahead] = (losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                      {k: (v["exp_avg"].cpu(), v["exp_avg_sq"].cpu()) for k, v in opt.state_dict()["state"].items()})
    assert res["0"][0] == res["1"][0]
    for k in res["0"][1]:
        assert torch.equal(res["0"][1][k], res["1"][1][k]), k
    for k in res["0"][2]:
        assert torch.equal(res["0"][2][k][0], res["1"][2][k][0]) and torch.equal(res["0"][2][k][1], res["1"][2][k][1]), k


@pytest.mark.parametrize("use_graph,sort_ahead,precision,emb_dim", [(False, True, "f32", 16), (True, True, "bf16", 16), (False, False, "bf16", 16),
                                                                     (False, True, "f32", 6)])
def test_rows_and_dense_parameters_updated_by_one_launch_train_bit_identically(cuda, use_graph, sort_ahead, precision, emb_dim):
    """cdc_embed_segsum_lazy_update_dense (the step's table rows and the dense parameters' Adam in ONE launch, TrainStep's default on
    one GPU with the lazy table) against cdc_adam_multi followed by cdc_embed_segsum_lazy_update: per element the same arithmetic,
    the split-K slabs added in the same order — weights, moments and the BCE losses held to BIT equality, the regularisation figure
    (double atomics in both forms) to 1e-6 relative.  A field with few distinct ids gives segments of every length class."""
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    from cdcmdr_amd.model.ple import PLE
    fd = [50, 3, 500, 7, 90, 4, 3000, 30]
    B, n = 1024, 6
    r = np.random.default_rng(18)
    Xs = [torch.from_numpy(make_ids(r, B, fd)).to(cuda) for _ in range(n + 1)]
    ys = [torch.from_numpy(r.integers(0, 2, size=B).astype(np.int16)).to(cuda) for _ in range(n)]
    gs = [torch.from_numpy(r.integers(0, 3, size=B).astype(np.int64)).to(cuda) for _ in range(n)]
    res = {}
    for one in (False, True):
        torch.manual_seed(4)
        # (emb_dim 6: rows without whole 16-byte pieces — the scalar forms of both update bodies)
        model = PLE(fd, emb_dim, 3, 2, 2, ((64, 32), (16,)), (8, 4), dropout=0.2).to(cuda).set_precision(precision)
        model.seed = 5
        opt = FusedAdam(model, table_mode="lazy", flush_every=4)
        ts = TrainStep(model, opt, B, use_graph=use_graph, sort_ahead=sort_ahead, rows_dense_one_launch=one)
        losses, regs = [], []
        for i in range(n):
            bce, reg = ts.step(Xs[i], ys[i], gs[i], next_X=Xs[i + 1])
            losses.append(float(bce.item()))
            regs.append(float(reg.item()))
        ts.check_ids()
        opt.flush_table()
        res[one] = (losses, regs, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                    {k: (v["exp_avg"].cpu(), v["exp_avg_sq"].cpu()) for k, v in opt.state_dict()["state"].items()})
    assert res[False][0] == res[True][0]
    assert np.allclose(res[False][1], res[True][1], rtol=1e-6, atol=0.0), (res[False][1], res[True][1])
    for k in res[False][2]:
        assert torch.equal(res[False][2][k], res[True][2][k]), k
    for k in res[False][3]:
        assert torch.equal(res[False][3][k][0], res[True][3][k][0]) and torch.equal(res[False][3][k][1], res[True][3][k][1]), k


def test_table_adam_kernels_bits_equal_the_c_restatement(cuda):
    """The dense streaming pass (untouched rows: L2-only gradient) and the touched-row kernel against
    oracle/adam_elem_ref.c, which tests/test_host_logic.py pins bit-for-bit to torch's CPU Adam."""
    import ctypes as C
    import math
    from test_host_logic import _build_adam_ref
    from cdcmdr_amd.model.dcn import DCN
    from cdcmdr_amd.optim import FusedAdam
    ref = _build_adam_ref()
    fd = [300, 50]
    torch.manual_seed(2)
    model = DCN(fd, 8, 1, (8,), dropout=0.0).to(cuda)
    opt = FusedAdam(model, table_mode="dense")
    w0 = opt.table.detach().cpu().numpy().copy()
    R, D = w0.shape
    B = 64
    rng = np.random.default_rng(0)
    idx = torch.from_numpy(np.stack([rng.integers(0, 300, size=B), 300 + rng.integers(0, 50, size=B)], axis=1).astype(np.int32)).to(cuda)
    dE = torch.randn(B, 2 * D, generator=torch.Generator().manual_seed(1)).to(cuda)
    w, m, v = w0.reshape(-1).copy(), np.zeros(R * D, np.float32), np.zeros(R * D, np.float32)
    f32 = lambda x: float(np.float32(x))  # noqa: E731
    for t in range(1, 4):
        opt.begin_step()
        opt.table_step(idx, dE, B, 2, D)
        g = np.zeros((R, D), np.float32)
        di = dE.cpu().numpy().reshape(B, 2, D)
        ii = idx.cpu().numpy()
        for b in range(B):                                    # ascending batch order, like aten::embedding_dense_backward
            for f in range(2):
                g[ii[b, f]] += di[b, f]
        gf = g.reshape(-1)
        ref.adam_elem_ref(w.ctypes.data, m.ctypes.data, v.ctypes.data, gf.ctypes.data, R * D, f32(1 - 0.9), f32(0.99), f32(1 - 0.99),
                          f32(1e-8), f32(1e-8), 2 * f32(1e-5), f32(1e-3 / (1 - 0.9 ** t)), f32(math.sqrt(1 - 0.99 ** t)))
        gw = opt.table.detach().cpu().numpy().reshape(-1)
        gm, gv = opt.table_m.cpu().numpy().reshape(-1), opt.table_v.cpu().numpy().reshape(-1)
        # m and v: bit-identical but for a handful; w: the device's fp32 divide/sqrt sequence differs from the host libm
        # in the last bit of the update quotient for ~0.2 % of the elements, which flips w's rounding there
        for name, got, want, floor in (("w", gw, w, 0.99), ("m", gm, m, 0.995), ("v", gv, v, 0.995)):
            same = float((got == want).mean())
            ulp = int(np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)).max())
            assert same >= floor and ulp <= 8, f"step {t} table {name}: {same:.4f} bit-identical, max {ulp} ulp"
        w, m, v = gw.copy(), gm.copy(), gv.copy()             # continue from the device state (no drift accumulation)


def test_train_epoch_over_a_device_loader_with_a_ragged_last_batch(cuda):
    """Run.train's epoch loop (run.py:470-497) on the DeviceLoader — three batches of 32, 32 and 16 rows
    (the reference trains on the ragged tail like on any other batch) against the CPU oracle driven by torch's own Adam."""
    from cdcmdr_amd.data import make_loader, train_epoch
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = FD_SPARSE
    torch.manual_seed(21)
    model = PLE(fd, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0).to(cuda).set_precision("f32")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(8)
    n = 80
    X = torch.from_numpy(make_ids(rng, n, fd))
    X[:, 2] = torch.from_numpy(rng.integers(0, 3, size=n).astype(np.int32))
    y = torch.from_numpy(rng.integers(0, 2, size=(n, 1)).astype(np.int16))
    loader, w = make_loader(X, y, 32, cuda, domain_idx=2, domain2group={0: 0, 1: 1, 2: 2}, shuffle=False)
    opt = FusedAdam(model, table_mode="dense")          # dense: the logged regularisation value is exact every step
    ts = TrainStep(model, opt, 32)
    logged = []
    done, skipped = train_epoch(ts, loader, log_interval=1, log=logged.append)
    assert (done, skipped) == (3, 0) and len(logged) == 3
    opt.flush_table()
    # the oracle: same batches, torch.optim.Adam on the CPU (what run.py:481-493 makes torch do)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    s2 = dict(sd)
    s2.update(leaves)
    l2 = {k: 1e-5 for k in O.reg_names(list(sd), "ple")}
    ref_opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    names = set(sd)
    want_losses = []
    for lo in (0, 32, 64):
        xb, yb = X[lo:lo + 32].numpy(), y[lo:lo + 32, 0].float()
        gb = X[lo:lo + 32, 2].to(torch.int64).reshape(-1, 1)
        stats = {}
        p = O.ple_forward(s2, xb, fd, 3, training=True, stats_out=stats).gather(1, gb).squeeze(1)
        loss = O.bce_mean(p, yb) + O.reg_loss(s2, l2)
        ref_opt.zero_grad()
        loss.sum().backward()
        ref_opt.step()
        want_losses.append(float(loss.sum().detach()))
        s2.update(stats)
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for a, b in zip(logged, want_losses):
        assert abs(a - b) < 2e-5 * max(1.0, abs(b)), (logged, want_losses)
    for k in names:
        if is_pre_bn_bias(k, names) or "num_batches" in k:
            continue
        # a running mean contains the (rounding-noise driven, +-lr per step) bias of the Linear in front of it: up to
        # momentum * sum over the steps of the difference — 1e-3 after three steps (measured 4.9e-4)
        atol = 1.5e-3 if k.endswith("running_mean") else 2e-5
        assert_close(got[k], s2[k].detach(), 5e-4, atol, f"epoch: {k}")


def test_first_logged_loss_of_a_lazy_run_carries_the_tables_l2_term(cuda):
    """run.py:489 adds l2 * sum(w^2) over the WHOLE table to every reported loss.  The lazy table caches that term
    (TrainStep.refresh_table_reg); a freshly built step must have it from the very first logging window: the values
    train_epoch logs on the lazy table equal the dense-mode values, where the term is summed every step."""
    from cdcmdr_amd.data import make_loader, train_epoch
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = FD_SPARSE
    rng = np.random.default_rng(8)
    n = 96
    X = torch.from_numpy(make_ids(rng, n, fd))
    X[:, 2] = torch.from_numpy(rng.integers(0, 3, size=n).astype(np.int32))
    y = torch.from_numpy(rng.integers(0, 2, size=(n, 1)).astype(np.int16))
    logged = {}
    for table_mode in ("dense", "lazy"):
        torch.manual_seed(21)
        model = PLE(fd, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0).to(cuda).set_precision("f32")
        loader, _ = make_loader(X, y, 32, cuda, domain_idx=2, domain2group={0: 0, 1: 1, 2: 2}, shuffle=False)
        opt = FusedAdam(model, table_mode=table_mode, fast_replay=False)
        ts = TrainStep(model, opt, 32)
        out = []
        train_epoch(ts, loader, log_interval=1, log=out.append)
        logged[table_mode] = out
        # and without train_epoch: the very first step() of a new TrainStep reports the term as well
        torch.manual_seed(21)
        model2 = PLE(fd, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0).to(cuda).set_precision("f32")
        ts2 = TrainStep(model2, FusedAdam(model2, table_mode=table_mode, fast_replay=False), 32)
        bce, reg = ts2.step(X[:32].to(cuda), y[:32, 0].to(cuda), X[:32, 2].to(torch.int64).to(cuda))
        logged[table_mode + "_first"] = float(bce.item()) + float(reg.item())
    table_term = 1e-5 * float((model.embedding.embedding_dict.weight.detach().double() ** 2).sum())
    assert table_term > 1e-3                                        # a term that would be missed, not a rounding matter
    for a, b in zip(logged["dense"], logged["lazy"]):
        assert abs(a - b) < 1e-6 * max(1.0, abs(a)), (logged["dense"], logged["lazy"])
    assert abs(logged["dense_first"] - logged["lazy_first"]) < 1e-6 * max(1.0, abs(logged["dense_first"]))
    assert abs(logged["dense"][0] - logged["dense_first"]) < 1e-9


def test_star_fast_path_matches_the_oracle(cuda):
    """TrainStep(mode="star"): rows partitioned by domain, every partition through its own tower + the shared one
    (star.py:109-181), BCE on the group-ordered targets (run.py:476-479) — two steps against the oracle's grouped forward
    driven by torch autograd + torch.optim.Adam; one of the five domains has no row in the second batch."""
    from cdcmdr_amd.model.star import STAR
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = [7, 400, 3, 50, 5, 29]
    torch.manual_seed(9)
    model = STAR(fd, 4, 5, (32, 16, 8), domain_idx=4, dropout=0.0).to(cuda).set_precision("f32")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = FusedAdam(model, table_mode="dense")
    ts = TrainStep(model, opt, 96, mode="star")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    s2 = dict(sd)
    s2.update(leaves)
    l2 = {k: 1e-5 for k in O.reg_names(list(sd), "star")}
    ref_opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    rng = np.random.default_rng(6)
    names = set(sd)
    for step in range(2):
        X = make_ids(rng, 96, fd)
        if step == 1:
            X[X[:, 4] == 3, 4] = 2                                  # domain 3 absent
        y = rng.integers(0, 2, size=96).astype(np.int16)
        g = X[:, 4].astype(np.int64)
        bce, _ = ts.step(torch.from_numpy(X).to(cuda), torch.from_numpy(y).to(cuda), torch.from_numpy(g).to(cuda))
        stats = {}
        p, t = O.star_forward(s2, X, fd, 5, x_group=g, targets=torch.from_numpy(y).float(), training=True, stats_out=stats)
        loss = O.bce_mean(p.squeeze(1), t) + O.reg_loss(s2, l2)
        ref_opt.zero_grad()
        loss.sum().backward()
        if step == 0:
            used = {k for k, leaf in leaves.items() if leaf.grad is not None}      # every domain is present in batch 0
        for k, leaf in leaves.items():
            # the reference runs an absent domain's tower on an empty tensor: its parameters get exact ZERO gradients
            # (tests/golden/g4), so Adam still moves them (momentum, weight decay) — the oracle's forward skips the tower.
            # Parameters the forward never touches (shared_dnn's own BatchNorm) keep grad None on both sides.
            if leaf.grad is None and k in used:
                leaf.grad = torch.zeros_like(leaf)
        ref_opt.step()
        s2.update(stats)
        assert abs(float(bce.item()) - float(O.bce_mean(p.squeeze(1), t).detach())) < 2e-5
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for k in names:
        if "num_batches" in k or is_pre_bn_bias(k, names):
            continue
        if k == "shared_bn_bias" or (k.startswith("domain_norm.") and k.endswith(".bias")):
            continue        # a constant added in front of Linear -> BatchNorm: zero gradient up to rounding noise, like a pre-BN bias
        atol = 5e-4 if k.endswith("running_mean") else 2e-5
        assert_close(got[k], s2[k].detach(), 5e-4, atol, f"star step: {k}")


def test_attention_model_trains_on_the_fast_path_graph_equals_eager(cuda):
    """PLE with the attention branch and dropout 0.2 (dropout on the attention probabilities included) through TrainStep:
    the hipGraph replay reproduces the eager run bit for bit (the dropout streams are keyed by the device step counter),
    the loss goes down, every attention parameter moves."""
    import types
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = [20, 500, 7, 90, 4]
    cfg = types.SimpleNamespace(use_atten=True, atten_embed_dim=16, att_layer_num=2, att_head_num=2, att_res=True, use_dcn=False)
    res = {}
    for use_graph in (False, True):
        torch.manual_seed(4)
        model = PLE(fd, 8, 3, 1, 1, ((16,), (8,)), (8,), 0.2, cfg).to(cuda).set_precision("f32")
        w0 = {k: v.detach().clone() for k, v in model.named_parameters() if "atten" in k or "self_attns" in k or "V_res" in k}
        opt = FusedAdam(model, table_mode="lazy")
        ts = TrainStep(model, opt, 64, use_graph=use_graph)
        r = np.random.default_rng(8)
        X = torch.from_numpy(make_ids(r, 64, fd)).to(cuda)
        y = torch.from_numpy((make_ids(r, 64, [2])[:, 0]).astype(np.int16)).to(cuda)
        g = torch.from_numpy(r.integers(0, 3, size=64).astype(np.int64)).to(cuda)
        losses = []
        for _ in range(30):                                        # the same batch: the loss must fall
            bce, _ = ts.step(X, y, g)
            losses.append(float(bce.item()))
        opt.flush_table()
        res[use_graph] = (losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        assert losses[-1] < losses[0] - 0.05, losses
        assert len(w0) >= 9 and all(float((dict(model.named_parameters())[k] - v).abs().max()) > 0 for k, v in w0.items())
    assert res[False][0] == res[True][0]
    for k in res[False][1]:
        assert torch.equal(res[False][1][k], res[True][1][k]), k


@pytest.mark.parametrize("table_mode", ["dense", "lazy"])
def test_checkpoint_resume_continues_bit_identically(cuda, tmp_path, table_mode):
    """run.py:447-449 saves {'state_dict', 'optimizer'}: three steps, a checkpoint through torch.save / torch.load
    (weights_only), a fresh model + optimiser + step restored from it, three more steps == six uninterrupted steps."""
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = [50, 3000, 7, 900]

    def make():
        torch.manual_seed(11)
        model = PLE(fd, 8, 3, 1, 1, ((16,), (8,)), (8,), dropout=0.2).to(cuda).set_precision("f32")
        opt = FusedAdam(model, table_mode=table_mode, fast_replay=False, flush_every=4)
        return model, opt

    r = np.random.default_rng(5)
    batches = [(torch.from_numpy(make_ids(r, 128, fd)).to(cuda), torch.from_numpy(r.integers(0, 2, size=128).astype(np.int16)).to(cuda),
                torch.from_numpy(r.integers(0, 3, size=128).astype(np.int64)).to(cuda)) for _ in range(6)]
    model, opt = make()
    ts = TrainStep(model, opt, 128)
    for b in batches:
        ts.step(*b)
    opt.flush_table()
    want = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    want_m = opt.table_m.cpu().clone()

    model, opt = make()
    ts = TrainStep(model, opt, 128)
    for b in batches[:3]:
        ts.step(*b)
    path = os.path.join(tmp_path, "ckpt.pth.tar")
    torch.save({"state_dict": model.state_dict(), "optimizer": opt.state_dict()}, path)
    del ts, opt, model
    ck = torch.load(path, map_location=cuda, weights_only=True)
    model, opt = make()
    model.load_state_dict(ck["state_dict"])
    opt.load_state_dict(ck["optimizer"])
    assert int(opt.step_dev.item()) == 3
    ts = TrainStep(model, opt, 128)
    for b in batches[3:]:
        ts.step(*b)
    opt.flush_table()
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), want[k]), f"{k} differs after resume"
    assert torch.equal(opt.table_m.cpu(), want_m)


@pytest.mark.parametrize("use_graph,ahead", [(False, True), (True, True), (False, False)])
def test_an_id_that_aliases_another_fields_row_is_reported(cuda, use_graph, ahead):
    """An id outside its field's vocabulary that still lies INSIDE the table (field 0 has 50 rows: id 50 is row 0 of field 1).  The
    reference gathers that row like any other (model/layer.py:152-153) — so does the gather here, bit for bit — and its dense
    backward sums the gradients of both fields into ONE Adam update; the per-field row lists of this path would update the row once
    per field.  Instead of diverging silently the step reports it: check_ids() raises ValueError naming the position (an id that
    leaves the table altogether keeps raising IndexError), and a clean batch afterwards trains normally."""
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = [50, 3000, 3, 900, 7]
    B, D = 192, 8
    r = np.random.default_rng(3)
    X = make_ids(r, B, fd)
    y = torch.from_numpy(r.integers(0, 2, size=B).astype(np.int16)).to(cuda)
    g = torch.from_numpy(X[:, 2].astype(np.int64)).to(cuda)
    good = torch.from_numpy(X).to(cuda)
    alias = good.clone()
    alias[5, 0] = 50 + 17                                               # field 0: outside [0, 50), row 17 of field 1
    torch.manual_seed(4)
    model = PLE(fd, D, 3, 1, 1, ((16,), (8,)), (8,), dropout=0.0).to(cuda).set_precision("f32")
    opt = FusedAdam(model, table_mode="lazy", flush_every=4)
    ts = TrainStep(model, opt, B, use_graph=use_graph, sort_ahead=ahead)
    for _ in range(3):
        ts.step(good, y, g, next_X=good)
    ts.check_ids()
    table = model.embedding.embedding_dict.weight
    ts.step(alias, y, g, next_X=good)
    opt.flush_table()
    got = ts.emb.out.tensor()[5, 0:D].clone()
    with pytest.raises(ValueError, match="batch position 5, field 0"):
        ts.check_ids()
    ts.step(good, y, g, next_X=good)
    ts.check_ids()                                                       # the flag was cleared, a clean batch passes
    assert np.isfinite(float(ts.loss.item()))
    # the gather itself took the aliased row, as the reference does (checked on a fresh forward through the drop-in path)
    model.eval()
    with torch.no_grad():
        model(alias)
    emb = model.plan_holder(B).emb_op.out.tensor()
    assert torch.equal(emb[5, 0:D], table.detach()[50 + 17])
    assert got.shape == (D,)
    # out of the table altogether: IndexError first
    worse = alias.clone()
    worse[9, 4] = 7
    ts.step(worse, y, g)
    with pytest.raises(IndexError):
        ts.check_ids()
    ts.step(good, y, g)
    ts.check_ids()


@pytest.mark.parametrize("D,precision", [(16, "bf16"), (32, "f32"), (8, "f32")])
def test_fused_catchup_gather_equals_the_two_launches(cuda, monkeypatch, D, precision):
    """cdc_embed_lazy_catchup_gather (the catch-up lanes write the embeddings) against cdc_embed_lazy_catchup followed by
    cdc_embed_gather_fwd: identical embeddings, losses and table bits over several steps — with a hot column (three rows looked up
    by a third of the batch each: the wave-cooperative path), rows looked up 2..4 times (inline path), repeated batches (rows
    that are already current) and an id outside the table (zero row + IndexError from check_ids)."""
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = [50, 3000, 3, 900, 7]
    B = 192
    r = np.random.default_rng(9)
    batches = []
    for _ in range(5):
        X = make_ids(r, B, fd)
        batches.append((torch.from_numpy(X).to(cuda), torch.from_numpy(r.integers(0, 2, size=B).astype(np.int16)).to(cuda),
                        torch.from_numpy(X[:, 2].astype(np.int64)).to(cuda)))
    batches.append(batches[1])                                          # the same rows again, two steps later
    bad = batches[2][0].clone()
    # (an id past its field's vocabulary that still lies INSIDE the table: test_an_id_that_aliases_another_fields_row_is_reported)
    bad[7, 4] = 7                                                       # last field: 7 is outside [0, 7) and outside the table
    res = {}
    for fused in ("1", "0"):
        torch.manual_seed(4)
        model = PLE(fd, D, 3, 1, 1, ((16,), (8,)), (8,), dropout=0.0).to(cuda).set_precision(precision)
        opt = FusedAdam(model, table_mode="lazy", flush_every=4)
        ts = TrainStep(model, opt, B, use_graph=False, fuse_gather=(fused == "1"))
        assert ts._fuse_gather() == (fused == "1")
        losses, embs = [], []
        for b in batches:
            bce, _ = ts.step(*b)
            losses.append(float(bce.item()))
            embs.append(ts.emb.out.tensor().clone())
        ts.step(bad, batches[2][1], batches[2][2])
        embs.append(ts.emb.out.tensor().clone())
        with pytest.raises(IndexError):
            ts.check_ids()
        opt.flush_table()
        res[fused] = (losses, embs, model.embedding.embedding_dict.weight.detach().clone(), opt.table_m.clone(), opt.table_v.clone())
    assert res["1"][0] == res["0"][0]
    for a, b in zip(res["1"][1], res["0"][1]):
        assert torch.equal(a, b), "gathered embeddings differ"
    assert float(res["1"][1][-1][7, 4 * D:5 * D].abs().max()) == 0.0, "an id outside the table must give a zero row"
    for k in (2, 3, 4):
        assert torch.equal(res["1"][k], res["0"][k]), "table state differs"


def test_load_state_dict_into_a_graph_captured_step(cuda):
    """A captured hipGraph holds the moment POINTERS of cdc_adam_multi's argument blocks: load_state_dict must restore into the
    existing tensors.  Three steps (the third replayed from the graph), a checkpoint, three more steps, then the checkpoint
    loaded back INTO THE SAME warmed objects and the three steps repeated == a fresh model + optimiser resumed from the same
    checkpoint, bit for bit (what Runner.fit does when it reloads the best checkpoint into its live TrainStep)."""
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    fd = [50, 3000, 7, 900]

    def make():
        torch.manual_seed(11)
        model = PLE(fd, 8, 3, 1, 1, ((16,), (8,)), (8,), dropout=0.0).to(cuda).set_precision("f32")
        return model, FusedAdam(model, table_mode="lazy", fast_replay=False, flush_every=4)

    r = np.random.default_rng(5)
    batches = [(torch.from_numpy(make_ids(r, 128, fd)).to(cuda), torch.from_numpy(r.integers(0, 2, size=128).astype(np.int16)).to(cuda),
                torch.from_numpy(r.integers(0, 3, size=128).astype(np.int64)).to(cuda)) for _ in range(6)]
    model, opt = make()
    ts = TrainStep(model, opt, 128, use_graph=True)
    for b in batches[:3]:
        ts.step(*b)
    assert ts.graph is not None                                          # step 3 was captured and replayed
    ck = {"state_dict": {k: v.detach().clone() for k, v in model.state_dict().items()}, "optimizer": opt.state_dict()}
    ptrs = {k: (m.data_ptr(), v.data_ptr()) for k, (m, v) in opt.state.items()}
    for b in batches[3:]:
        ts.step(*b)                                                      # moves weights and moments away from the checkpoint
    model.load_state_dict(ck["state_dict"])
    opt.load_state_dict(ck["optimizer"])
    assert {k: (m.data_ptr(), v.data_ptr()) for k, (m, v) in opt.state.items()} == ptrs, "moment tensors were re-allocated"
    for b in batches[3:]:
        ts.step(*b)                                                      # graph replays on the restored state
    opt.flush_table()
    got = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    got_m = {k: m.cpu().clone() for k, (m, _) in opt.state.items()}
    names = {id(p): n for n, p in model.named_parameters()}
    got_m = {names[k]: v for k, v in got_m.items()}

    model2, opt2 = make()
    model2.load_state_dict(ck["state_dict"])
    opt2.load_state_dict(ck["optimizer"])
    ts2 = TrainStep(model2, opt2, 128)
    for b in batches[3:]:
        ts2.step(*b)
    opt2.flush_table()
    names2 = {id(p): n for n, p in model2.named_parameters()}
    for k, v in model2.state_dict().items():
        assert torch.equal(v.cpu(), got[k]), f"{k} differs between the in-place restore and a fresh resume"
    for k, (m, _) in opt2.state.items():
        assert torch.equal(m.cpu(), got_m[names2[k]]), f"exp_avg of {names2[k]} differs"


def test_runner_fit_keeps_the_best_checkpoint_and_stops_early(cuda, tmp_path):
    """Run.main / is_continuable (run.py:440-468, 713-770) on the HIP path: epochs of training, validation through the
    device-side evaluator, best checkpoint by mean_auc, early stop, reload of the best model, test-set evaluation."""
    from cdcmdr_amd.data import make_loader
    from cdcmdr_amd.evaluate import Evaluator
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.runner import Runner
    from cdcmdr_amd.trainer import TrainStep
    fd = [20, 400, 3, 60, 9]
    rng = np.random.default_rng(3)
    n = 4096
    X = torch.from_numpy(make_ids(rng, n, fd))
    w = rng.standard_normal(60)
    y = torch.from_numpy(((w[X[:, 3].numpy()] + 0.5 * rng.standard_normal(n)) > 0).astype(np.int16)).reshape(-1, 1)   # learnable labels
    d2g = {0: 0, 1: 1, 2: 2}
    tr, wgt = make_loader(X[:3072], y[:3072], 256, cuda, domain_idx=2, domain2group=d2g)
    va, _ = make_loader(X[3072:3584], y[3072:3584], 256, cuda, domain_idx=2, domain2group=d2g, shuffle=False)
    te, _ = make_loader(X[3584:], y[3584:], 256, cuda, domain_idx=2, domain2group=d2g, shuffle=False)
    torch.manual_seed(0)
    model = MMoE(fd, 8, 3, 4, (32, 16), (8,), dropout=0.1).to(cuda).set_precision("f32")
    opt = FusedAdam(model, table_mode="lazy")
    ts = TrainStep(model, opt, 256, use_graph=True)
    ev = Evaluator(model, mode="multi", domain_idx=2, n_domain=3, domain_cnt_weight=wgt)
    logs = []
    runner = Runner(model, ts, ev, os.path.join(tmp_path, "best.pth.tar"), num_trials=2, log=logs.append)
    out = runner.fit(tr, va, te, epochs=6)
    assert 1 <= out["epochs_run"] <= 6 and os.path.exists(os.path.join(tmp_path, "best.pth.tar"))
    assert out["best"]["total_auc"] > 0.6 and out["test"]["total_auc"] > 0.6           # it learned the planted signal
    assert runner.best_mean_auc == out["best"]["mean_auc"] and set(out["test"]["domain_auc"]) == {0, 1, 2}
    # the model now holds the best checkpoint's weights: re-evaluating the validation set reproduces its figures
    again = ev.test(va)
    assert abs(again["total_auc"] - out["best"]["total_auc"]) < 1e-12 and abs(again["mean_loss"] - out["best"]["mean_loss"]) < 1e-12
