"""The fused tower launches (csrc/tower.hip: Linear -> BatchNorm -> ReLU -> Dropout -> Linear -> BatchNorm -> ReLU -> Dropout ->
Linear(->1) + wide term + Sigmoid of all towers in ONE launch per direction, BatchNorm statistics exchanged between workgroups
inside the launch) against the launches they replace — cdc_gemm_bf16_nt -> cdc_bn_fwd -> cdc_gemm_bf16_nt -> cdc_bn_fwd ->
cdc_head_fwd and the mirror chain — and against the oracle (model/layer.py:35-56,178-206 of the reference).

Forward: the contractions, the statistics sums (per 64-row chunk, chunks in cdc_bn_fwd's order) and the dropout stream are those of
the unfused chain, so every saved tensor up to the second hidden activation is held to BIT equality; the head's dot product over
the wide input uses 16-byte lanes (another summation order), so probabilities are held to 2e-6.  Backward: column sums per
128-row block instead of k_bn_bwd_stats_v4's shuffle tree: the towers' gradients within 2e-4 relative L2 of the unfused chain,
the gradient handed to the level below within 2e-5."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import O, assert_close, compare_param_grads, is_pre_bn_bias, make_ids, oracle_grads, sd_cpu

pytestmark = pytest.mark.gpu

FD = [1000] * 26
DIMS, TOWER, D = ((256, 128), (64,)), (64, 32), 16


def _model(cuda, n_tower, dropout, fused, seed=0, kind="ple"):
    from cdcmdr_amd import plan as P
    torch.manual_seed(seed)
    if kind == "ple":
        from cdcmdr_amd.model.ple import PLE
        m = PLE(FD, D, n_tower, 2, 2, DIMS, TOWER, dropout=dropout)
    else:                                                      # MMoE: the towers read the 128-wide expert mixture (H0 = 128)
        from cdcmdr_amd.model.mmoe import MMoE
        m = MMoE(FD, D, n_tower, 4, (256, 128), TOWER, dropout=dropout)
    m = m.to(cuda).set_precision("bf16")
    m.seed = 1234
    P.TowerChain.enabled = fused
    return m


@pytest.fixture(autouse=True)
def _restore_switch():
    from cdcmdr_amd import plan as P
    yield
    P.TowerChain.enabled = True


def _chain(m, B, tag=None):
    from cdcmdr_amd import plan as P
    holder = m.plan_holder(B)
    hits = [op for op in holder.plan.ops if isinstance(op, P.TowerChain)]
    return holder, (hits[0] if hits else None)


def _tower_bufs(holder, chain):
    """(z1, a1 shadow, z2, a2) tensors of the plan, fused or not"""
    from cdcmdr_amd import plan as P
    plan = holder.plan
    if chain is not None:
        l1, b1, l2, b2 = chain.l1, chain.b1, chain.l2, chain.b2
    else:
        ops = plan.ops
        i = max(k for k, op in enumerate(ops) if isinstance(op, P.TowerHead))
        l1, b1, l2, b2 = ops[i - 4:i]
    z1 = l1.groups[0]["y"].root
    a1 = plan._shadow_root(b1.segs[0]["y"].root)
    z2 = l2.groups[0]["y"].root
    a2 = b2.segs[0]["y"].root
    return z1, a1, z2, a2


@pytest.mark.parametrize("kind,n_tower,B,dropout", [("ple", 3, 4096, 0.2), ("ple", 3, 1000, 0.2), ("ple", 4, 300, 0.0), ("ple", 3, 130, 0.2),
                                                    ("ple", 3, 2, 0.0), ("ple", 2, 8192, 0.2), ("mmoe", 3, 1000, 0.2)])
def test_fused_towers_equal_the_five_launches(cuda, kind, n_tower, B, dropout):
    from cdcmdr_amd import plan as P
    rng = np.random.default_rng(B)
    x = torch.from_numpy(make_ids(rng, B, FD)).to(cuda)
    gout = torch.randn((B, n_tower), generator=torch.Generator().manual_seed(7)).to(cuda)
    res = {}
    for fused in (False, True):
        m = _model(cuda, n_tower, dropout, fused, kind=kind)
        m.train()
        out = m(x)
        holder, chain = _chain(m, B)
        assert (chain is not None) == fused
        saved = [t.detach().clone() for t in _tower_bufs(holder, chain)]
        out.backward(gout)
        torch.cuda.synchronize()
        if chain is not None:
            assert int(chain.tmo_word.item()) == 0
        stats = {k: v.detach().clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
        l1 = chain.l1 if chain is not None else [op for op in holder.plan.ops if isinstance(op, P.GLinear)][-2]
        res[fused] = (out.detach().clone(), saved, {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, stats,
                      holder.emb_op.out.grad.tensor().detach().clone(), torch.cat([g["x"].grad.tensor() for g in l1.groups], 1).detach().clone())
    out_u, saved_u, g_u, st_u, dE_u, dX_u = res[False]
    out_f, saved_f, g_f, st_f, dE_f, dX_f = res[True]
    for name, a, b in zip(("z1", "a1 (bf16)", "z2", "a2"), saved_u, saved_f):
        rows = B
        a, b = a[:rows], b[:rows]
        if name.startswith("a1"):
            a, b = a[:, :n_tower * 64], b[:, :n_tower * 64]
        assert torch.equal(a, b), f"{name}: fused forward differs from the unfused chain: max |d| {float((a.float() - b.float()).abs().max()):.3e}"
    for k in st_u:
        assert torch.equal(st_u[k], st_f[k]), f"{k} differs"
    assert_close(out_f, out_u, 0.0, 2e-6, "probabilities")
    assert set(g_u) == set(g_f)
    names = set(sd_cpu(m))
    worst = 0.0
    for k in g_u:
        a, b = g_u[k].double(), g_f[k].double()
        if is_pre_bn_bias(k, names):                           # mathematically zero: rounding noise on both sides
            assert float(b.abs().max()) <= 1e-1 * max(float(g_u[k[:-5] + ".weight"].abs().max()), 1e-3) + 1e-4, k
            continue
        rel = float((a - b).norm() / max(float(a.norm()), 1e-30))
        worst = max(worst, rel)
        # the towers' own gradients come straight out of the fused launch (measured <= 4e-5: a bf16 rounding of dZ that falls the
        # other way here and there); everything upstream passes through further bf16 roundings of the unchanged launches, where
        # each of those flips moves a whole row (measured <= 2.2e-4 at the first expert layer)
        bound = 2e-4 if k.startswith(("towers.", "linear.")) else 1e-3
        assert rel < bound or float((a - b).abs().max()) < 1e-7, f"{k}: fused vs unfused relative L2 {rel:.3e}"
    rel = float((dE_u.double() - dE_f.double()).norm() / dE_u.double().norm())
    assert rel < 1e-3, f"embedding gradient: relative L2 {rel:.3e}"
    relx = float((dX_u.double() - dX_f.double()).norm() / dX_u.double().norm())
    assert relx < 2e-5, f"gradient w.r.t. the towers' inputs: relative L2 {relx:.3e}"
    print(f"{kind} n_tower {n_tower} B {B}: saved forward tensors bit-equal; worst gradient relative L2 fused vs unfused {worst:.2e}, dE {rel:.2e}, dX {relx:.2e}")


def test_fused_towers_against_the_oracle(cuda):
    B = 512
    m = _model(cuda, 3, 0.0, True)
    m.train()
    rng = np.random.default_rng(3)
    x = make_ids(rng, B, FD)
    sd = sd_cpu(m)
    out = m(torch.from_numpy(x).to(cuda))
    assert _chain(m, B)[1] is not None
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
    out.backward(gout.to(cuda))
    O.MATMUL_BF16 = "exact"
    try:
        stats = {}
        ref, grads = oracle_grads(lambda s: O.ple_forward(s, x, FD, 3, training=True, stats_out=stats), sd, gout)
    finally:
        O.MATMUL_BF16 = False
    assert_close(out, ref, 5e-3, 2e-3, "probabilities")
    compare_param_grads(dict(m.named_parameters()), grads, 5e-3, 2e-3, bf16=True, all_names=list(sd))
    after = sd_cpu(m)
    for k, v in stats.items():                                 # running statistics of the towers' BatchNorms after the step
        if k.startswith("towers."):
            assert_close(after[k], v, 2e-3, 2e-4, k)


@pytest.mark.parametrize("table_mode,use_graph", [("dense", False), ("lazy", True)])
def test_training_steps_with_the_fused_towers_follow_the_unfused_trajectory(cuda, table_mode, use_graph):
    """the fused loss (BCE inside the backward launch), the head / wide / BatchNorm parameter gradients and the dense Adam that
    consumes them: four steps of TrainStep, fused towers against the five launches"""
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    B, n_tower, steps = 1024, 3, 4
    rng = np.random.default_rng(5)
    X = [torch.from_numpy(make_ids(rng, B, FD)).to(cuda) for _ in range(steps)]
    for x in X:
        x[:, 10] = x[:, 10] % n_tower
    y = [torch.from_numpy(rng.integers(0, 2, B).astype(np.int16)).to(cuda) for _ in range(steps)]
    res = {}
    for fused in (False, True):
        m = _model(cuda, n_tower, 0.2, fused)
        m.train()
        opt = FusedAdam(m, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, table_mode=table_mode)
        ts = TrainStep(m, opt, B, mode="multi", use_graph=use_graph)
        from cdcmdr_amd import plan as P
        assert any(isinstance(op, P.TowerChain) for op in ts.plan.ops) == fused
        losses = []
        for s in range(steps):
            g = X[s][:, 10].long()
            ts.step(X[s], y[s], g)
            losses.append(float(ts.loss.item()))
        ts.check_ids()
        if table_mode == "lazy":
            opt.flush_table()
        res[fused] = (losses, sd_cpu(m))
    lu, sdu = res[False]
    lf, sdf = res[True]
    assert abs(lu[0] - lf[0]) <= 2e-6, (lu, lf)
    for a, b in zip(lu, lf):
        assert abs(a - b) <= 1e-3 * max(abs(a), 1.0), (lu, lf)          # (step 1: the same bits; then Adam's +-lr moves part the runs)
    for k in sdu:
        if not sdu[k].dtype.is_floating_point:
            assert torch.equal(sdu[k], sdf[k]), k
            continue
        # Adam turns rounding-level differences of tiny gradients into +-lr moves: hold the trajectories to a few lr; the bias of a
        # Linear under a BatchNorm has a mathematically zero gradient, both runs move it by +-lr per step on rounding noise alone
        assert_close(sdf[k], sdu[k], 0.0, 2.1e-3 * steps if is_pre_bn_bias(k, set(sdu)) else 4.5e-3, k)
        d = (sdf[k].double() - sdu[k].double()).abs()
        if d.numel() >= 1024:                                  # (a four-element gate bias is one +-lr move away from any mean bound)
            assert float(d.mean()) < 2e-4, f"{k}: mean |d| {float(d.mean()):.3e}"


@pytest.mark.parametrize("B,n_tower,table_mode,use_graph,kind", [(1024, 3, "lazy", True, "ple"), (130, 3, "dense", False, "ple"),
                                                               (2, 3, "dense", False, "ple"), (300, 4, "dense", False, "ple"),
                                                               (1000, 3, "lazy", False, "mmoe")])
def test_both_directions_in_one_launch_equal_the_two_launches(cuda, B, n_tower, table_mode, use_graph, kind):
    """cdc_tower_step (TrainStep's default with the fused loss on one GPU) against cdc_tower_fwd + cdc_tower_bwd: the same arithmetic
    in the same order (the backward body reads what the forward body of the same workgroup wrote; the batch statistics travel in
    LDS instead of through save_mean / save_invstd) — losses and every parameter after four steps held to BIT equality."""
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    from cdcmdr_amd import plan as P
    steps = 4
    rng = np.random.default_rng(11)
    X = [torch.from_numpy(make_ids(rng, B, FD)).to(cuda) for _ in range(steps)]
    for x in X:
        x[:, 10] = x[:, 10] % n_tower
    y = [torch.from_numpy(rng.integers(0, 2, B).astype(np.int16)).to(cuda) for _ in range(steps)]
    res = {}
    for both in (False, True):
        m = _model(cuda, n_tower, 0.2, True, kind=kind)                # (MMoE: the towers read the 128-wide expert mixture, H0 = 128)
        m.train()
        opt = FusedAdam(m, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, table_mode=table_mode)
        ts = TrainStep(m, opt, B, mode="multi", use_graph=use_graph, tower_one_launch=both)
        chain = [op for op in ts.plan.ops if isinstance(op, P.TowerChain)][0]
        losses = []
        for s in range(steps):
            ts.step(X[s], y[s], X[s][:, 10].long())
            losses.append(float(ts.loss.item()))
            assert chain.one_launch is False                       # (only set around the launches of a step)
        ts.check_ids()
        assert (ts._tower_both is chain) == both
        if table_mode == "lazy":
            opt.flush_table()
        z1, a1, z2, a2 = _tower_bufs(m.plan_holder(B), chain)
        res[both] = (losses, sd_cpu(m), [t.detach().clone() for t in (z1, a1, z2, a2)])
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for u, v in zip(res[False][2], res[True][2]):
        assert torch.equal(u, v)
    for k, v in res[False][1].items():
        assert torch.equal(v, res[True][1][k]), k


@pytest.mark.parametrize("both", [True, False])
def test_a_wait_that_cannot_complete_gives_up_and_says_so(cuda, both):
    """The in-launch exchange is a bounded poll: with a record that is never published the launch still ends (~0.3 s),
    the error word carries CDC_TOWER_ERR_TIMEOUT, TrainStep.check_ids() raises, and the following step runs normally (the last workgroup
    of a launch puts the counters back)."""
    from cdcmdr_amd import _lib as L
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    from cdcmdr_amd import plan as P
    B, n_tower = 512, 3
    rng = np.random.default_rng(9)
    x = torch.from_numpy(make_ids(rng, B, FD)).to(cuda)
    x[:, 10] = x[:, 10] % n_tower
    y = torch.from_numpy(rng.integers(0, 2, B).astype(np.int16)).to(cuda)
    m = _model(cuda, n_tower, 0.0, True)
    m.train()
    opt = FusedAdam(m, table_mode="dense")
    ts = TrainStep(m, opt, B, mode="multi", use_graph=False, tower_one_launch=both)
    chain = [op for op in ts.plan.ops if isinstance(op, P.TowerChain)][0]
    ts.step(x, y, x[:, 10].long())
    ts.check_ids()
    good = float(ts.loss.item())
    # the test hook in the header of the workspace (128-byte lines of int32; csrc/tower.hip TW_MUTE): workgroup 1 of the next forward
    # publishes no layer-1 sums, so its tower's other workgroups never find them
    hdr = chain.ws.view(torch.int32)
    hdr[1 * 32] = 2
    ts.step(x, y, x[:, 10].long())
    torch.cuda.synchronize()
    assert int(chain.tmo_word.item()) & L.TOWER_ERR_TIMEOUT
    with pytest.raises(RuntimeError, match="timed out"):
        ts.check_ids()
    assert int(chain.ws.view(torch.int32)[:16 * 32].abs().sum().item()) == 0
    ts.step(x, y, x[:, 10].long())
    ts.check_ids()
    assert np.isfinite(float(ts.loss.item())) and abs(float(ts.loss.item()) - good) < 0.5
