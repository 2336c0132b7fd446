"""The data-parallel training step on the HIP path with two ranks sharing the one GPU of the test box (collectives go
through gloo with host staging here; on a multi-GPU node the same DataParallel calls are RCCL).  Checks (1) that the
replicas stay bit-identical (every rank applies the same table and dense updates from the exchanged global batch) in
both table modes, with the table replicated or row-sharded (ids / rows / row gradients exchanged with the owning rank by
all-to-all), and with the launch segments replayed as graphs, and (2) SURVEY.md §8e's parity definition: the 2-rank
step on shards == the 1-rank step on the concatenated batch (global-batch BatchNorm statistics, global-batch BCE mean)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FD = [30, 2000, 7, 300, 3]
B_LOCAL, STEPS = 64, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(world):
    rng = np.random.default_rng(11)
    n = B_LOCAL * world * STEPS
    X = np.stack([rng.integers(0, d, size=n) for d in FD], axis=1).astype(np.int32)
    y = rng.integers(0, 2, size=n).astype(np.int16)
    g = X[:, 4].astype(np.int64)
    if os.environ.get("CDC_TEST_GROUPS") == "cdc":                # C4: the tower of a row is the CLUSTER of its domain (30 -> 4)
        g = np.array([(3 * d + 1) % 4 for d in range(30)], dtype=np.int64)[X[:, 0]]
    return X, y, g


def _ragged_batch(world):
    """one more, RAGGED global batch (run.py:476 trains the tail of an epoch like any other batch): world * B_LOCAL - 5 rows"""
    rng = np.random.default_rng(23)
    n = B_LOCAL * world - 5
    X = np.stack([rng.integers(0, d, size=n) for d in FD], axis=1).astype(np.int32)
    return X, rng.integers(0, 2, size=n).astype(np.int16), X[:, 4].astype(np.int64)


def _build(kind, dev):
    D = int(os.environ.get("CDC_TEST_EMB_DIM", "8"))            # (spawned workers inherit the environment)
    if kind == "cdcple":
        # BASELINE config C4 in small: CDC(base='ple'), 30 domains (field 0) -> 4 clusters, emb_dim 32, nested expert dims
        import types
        from cdcmdr_amd.model.cdc import CDC
        cfg = types.SimpleNamespace(ple_n_expert_specific=2, ple_n_expert_shared=2, dataset_name="t", use_atten=False, use_dcn=False)
        cdc = CDC(FD, 32, 4, 30, "ple", ((32, 16), (8,)), (8, 4), 0, n_causal_mask=2, device=dev, dropout=0.0, config=cfg)
        return cdc.base_model_instance.to(dev).set_precision("f32"), "multi"
    if kind == "ple64":
        # PLE whose towers have the reference's widths (64 -> 64 -> 32 -> 1, config.py:39-42) in bf16: the towers run as the fused
        # launches (csrc/tower.hip) — split into six phases around the BatchNorm all-reduces under data parallelism
        from cdcmdr_amd.model.ple import PLE
        return PLE(FD, D, 3, 1, 1, ((32,), (64,)), (64, 32), dropout=0.0).to(dev).set_precision("bf16"), "multi"
    if kind == "star":
        from cdcmdr_amd.model.star import STAR
        return STAR(FD, D, 3, (32, 16), domain_idx=4, dropout=0.0).to(dev).set_precision("f32"), "star"
    from cdcmdr_amd.model.mmoe import MMoE
    return MMoE(FD, D, 3, 4, (32, 16), (8,), dropout=0.0).to(dev).set_precision("f32"), "multi"


def _single_process_reference(table_mode, kind="mmoe", world=2):
    """the same global batches through ONE rank (what the reference's single process would see)"""
    sys.path.insert(0, ROOT)
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    model, mode = _build(kind, dev)
    opt = FusedAdam(model, table_mode=table_mode, flush_every=2)
    gb = B_LOCAL * world
    ts = TrainStep(model, opt, gb, mode=mode)
    X, y, g = _data(world)
    losses = []
    for s in range(STEPS):
        sl = slice(s * gb, (s + 1) * gb)
        bce, _ = ts.step(torch.from_numpy(X[sl]).to(dev), torch.from_numpy(y[sl]).to(dev), torch.from_numpy(g[sl]).to(dev))
        losses.append(float(bce.item()))
    if os.environ.get("CDC_TEST_RAGGED") == "1":
        Xr, yr, gr = _ragged_batch(world)
        bce, _ = ts.sibling(Xr.shape[0]).step(torch.from_numpy(Xr).to(dev), torch.from_numpy(yr).to(dev), torch.from_numpy(gr).to(dev))
        losses.append(float(bce.item()))
    opt.flush_table()
    return {k: v.cpu() for k, v in model.state_dict().items()}, losses


def _worker(rank, world, port, out_dir, table_mode, use_graph, table_dist, sync_bn=True, kind="mmoe", backend="gloo", force=False):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    from cdcmdr_amd.dist import DataParallel
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    dp = DataParallel(backend=backend, force=force)
    torch.manual_seed(5)
    model, mode = _build(kind, dev)
    opt = FusedAdam(model, table_mode=table_mode, flush_every=2)
    if kind == "ple64":
        from cdcmdr_amd import plan as P
    ts = TrainStep(model, opt, B_LOCAL, mode=mode, use_graph=use_graph, dist=dp, table_dist=table_dist, sync_bn=sync_bn,
                   sort_ahead="force" if os.environ.get("CDC_TEST_AHEAD_FORCE") == "1" else True)
    assert ts.table_dist == (table_dist or ("sharded" if table_mode == "lazy" else "replicated"))
    X, y, g = _data(world)
    gb = B_LOCAL * world
    losses = []
    # CDC_TEST_AHEAD=1: every step names the next batch (step(..., next_X=)): local sort, bucketing and id exchange run one step
    # ahead; step 1 announces a batch that never comes (step 2 then sorts and exchanges its own), the last step announces nothing
    ahead = os.environ.get("CDC_TEST_AHEAD") == "1"
    shards = []
    for s in range(STEPS):
        lo = s * gb + rank * B_LOCAL
        sl = slice(lo, lo + B_LOCAL)
        shards.append((torch.from_numpy(X[sl]).to(dev), torch.from_numpy(y[sl]).to(dev), torch.from_numpy(g[sl]).to(dev)))
    decoy = shards[0][0].flip(0).contiguous()
    for s in range(STEPS):
        nxt = None
        if ahead and s + 1 < STEPS:
            nxt = decoy if s == 1 else shards[s + 1][0]
        bce, _ = ts.step(*shards[s], next_X=nxt)
        losses.append(float(bce.item()))
    if ahead and table_dist == "sharded" and table_mode == "lazy":
        assert ts._ahead_dp_ok and len(ts._dp_seqs) >= 3, "the look-ahead sequences were not used"
    if kind == "ple64":
        assert any(isinstance(op, P.TowerChain) for op in ts.plan.ops), "the fused tower launches were not chosen"
    if backend == "nccl" and use_graph:
        assert ts._one_graph_ok is True and ts._step_graphs, "the step was not captured as ONE graph with its collectives"
    if os.environ.get("CDC_TEST_RAGGED") == "1":
        # the epoch's ragged tail, split as data.DeviceLoader splits it: rank r gets base + (r < rem) rows
        Xr, yr, gr = _ragged_batch(world)
        n = Xr.shape[0]
        base, rem = divmod(n, world)
        a = rank * base + min(rank, rem)
        b = a + base + (1 if rank < rem else 0)
        sib = ts.sibling(b - a, global_rows=n, cap_rows=-(-n // world))
        bce, _ = sib.step(torch.from_numpy(Xr[a:b]).to(dev), torch.from_numpy(yr[a:b]).to(dev), torch.from_numpy(gr[a:b]).to(dev))
        losses.append(float(bce.item()))
        sib.check_ids()
    ts.check_ids()
    ts.gather_table()                 # flush + (row-sharded table) every owner's rows to every rank
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "m": opt.table_m.cpu(), "losses": losses},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dp.barrier()
    dp.close()


def test_two_ranks_row_sharded_table_with_emb_dim_32(cuda, tmp_path, monkeypatch):
    """C4's table shape (D = 32) through the row-sharded exchange: bucket, owner-side merge / catch-up / gather, expand, pack, owner
    update — replicas identical after gather_table() and equal to the single-process step on the concatenated batch."""
    monkeypatch.setenv("CDC_TEST_EMB_DIM", "32")
    test_two_ranks_stay_identical(cuda, tmp_path, "lazy", False, "sharded")


@pytest.mark.parametrize("table_mode,use_graph,table_dist", [("dense", False, None), ("lazy", False, "replicated"),
                                                             ("lazy", True, "replicated"), ("lazy", False, "sharded"),
                                                             ("lazy", True, "sharded")])
def test_two_ranks_stay_identical(cuda, tmp_path, table_mode, use_graph, table_dist):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), table_mode, use_graph, table_dist), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    assert r0["losses"] == r1["losses"]                              # the all-reduced global-batch loss
    assert all(np.isfinite(r0["losses"]))
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), f"replicas diverged in {k}"
    assert torch.equal(r0["m"], r1["m"])
    # N-rank step on shards == 1-rank step on the concatenated batch (fp32 summation order is all that differs)
    from helpers import assert_close, is_pre_bn_bias
    ref_sd, ref_losses = _single_process_reference(table_mode)
    for a, b in zip(r0["losses"], ref_losses):
        assert abs(a - b) < 2e-5, (r0["losses"], ref_losses)
    names = set(ref_sd)
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, names) or "num_batches" in k:
            continue                      # rounding-noise gradients that Adam turns into +-lr moves (tests/test_oracle_golden.py)
        # a running mean contains the (noise-driven) bias of the Linear in front of it: +-lr per step, times momentum
        atol = 5e-4 if k.endswith("running_mean") else 2e-5
        assert_close(r0["sd"][k], v, 5e-4, atol, f"2-rank vs 1-rank: {k}")
    # rows nobody looked up moved exactly like a single-process run's untouched rows (L2-only recurrence, STEPS steps)
    X, _, _ = _data(world)
    untouched = np.setdiff1d(np.arange(30, 2030), 30 + X[:, 1])
    w = r0["sd"]["embedding.embedding_dict.weight"][untouched]
    torch.manual_seed(5)
    sys.path.insert(0, ROOT)
    from cdcmdr_amd.model.mmoe import MMoE
    w0 = MMoE(FD, int(os.environ.get("CDC_TEST_EMB_DIM", "8")), 3, 4, (32, 16), (8,), dropout=0.0).state_dict()["embedding.embedding_dict.weight"][untouched]
    moved = (w - w0).abs()
    big = w0.abs() > 0.1
    assert float(moved[big].min()) > 0.9e-3 * STEPS and float(moved[big].max()) < 1.1e-3 * STEPS


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_train_the_ragged_last_batch_like_one_rank(cuda, tmp_path, monkeypatch, world):
    """The ragged last global batch of an epoch under data parallelism (run.py:476 trains it like any other batch; round 3 dropped
    it): world * 64 - 5 rows split unevenly over the ranks (data.DeviceLoader's split), trained through TrainStep.sibling(rows,
    global_rows=, cap_rows=) — BCE mean over the true global size, global BatchNorm statistics over unequal shares, row lists of a
    common capacity.  Replicas identical, and equal to ONE rank training the same rows as one ragged batch."""
    monkeypatch.setenv("CDC_TEST_RAGGED", "1")
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "lazy", False, "sharded"), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    assert len(rs[0]["losses"]) == STEPS + 1
    for r in rs[1:]:
        assert r["losses"] == rs[0]["losses"]
        for k in rs[0]["sd"]:
            assert torch.equal(rs[0]["sd"][k], r["sd"][k]), f"replicas diverged in {k}"
    from helpers import assert_close, is_pre_bn_bias
    ref_sd, ref_losses = _single_process_reference("lazy", world=world)
    for a, b in zip(rs[0]["losses"], ref_losses):
        assert abs(a - b) < 2e-5, (rs[0]["losses"], ref_losses)
    names = set(ref_sd)
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, names) or "num_batches" in k:
            continue
        atol = 6e-4 if k.endswith("running_mean") else 2e-5
        assert_close(rs[0]["sd"][k], v, 5e-4, atol, f"{world}-rank vs 1-rank with a ragged last batch: {k}")


@pytest.mark.parametrize("world,use_graph", [(2, False), (3, True)])
def test_ranks_with_the_fused_towers_split_around_the_batchnorm_exchanges(cuda, tmp_path, world, use_graph):
    """Global-batch BatchNorm statistics under data parallelism with the towers as fused launches (round 4): cdc_tower_dp runs the
    two launches of csrc/tower.hip as six phases, the local column sums of every BatchNorm all-reduced between them.  Replicas
    identical; and equal — up to the summation order of the statistics and what a bf16 rounding that falls the other way does over
    four Adam steps — to ONE rank running the monolithic launches (in-launch exchange) on the concatenated batches."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "lazy", use_graph, "sharded", True, "ple64"), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    for r in rs[1:]:
        assert r["losses"] == rs[0]["losses"] and all(np.isfinite(r["losses"]))
        for k in rs[0]["sd"]:
            assert torch.equal(rs[0]["sd"][k], r["sd"][k]), f"replicas diverged in {k}"
    from helpers import assert_close, is_pre_bn_bias
    ref_sd, ref_losses = _single_process_reference("lazy", kind="ple64", world=world)
    print("losses", rs[0]["losses"], ref_losses)
    for a, b in zip(rs[0]["losses"], ref_losses):
        assert abs(a - b) < 1e-5, (rs[0]["losses"], ref_losses)       # (measured: identical with 2 ranks, 6e-8 with 3)
    names = set(ref_sd)
    worst = 0.0
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, names) or "num_batches" in k:
            continue
        d = float((rs[0]["sd"][k].double() - v.double()).abs().max())
        worst = max(worst, d)
        assert d < 2e-4, f"{world}-rank split towers vs 1-rank monolithic towers: {k}: {d:.3e}"     # (measured 1e-7 / 6e-6)
    print(f"worst parameter difference after {STEPS} steps: {worst:.2e}")


def test_two_ranks_cdc_ple_row_sharded_equals_one_rank(cuda, tmp_path, monkeypatch):
    """BASELINE config C4's composition — CDC(base='ple') with 30 domains grouped into 4 cluster towers, emb_dim 32, trained in
    split mode on the lazy table, the table row-sharded over the ranks (ids / rows / row gradients by all-to-all): two ranks on
    shards == one rank on the concatenated batches, replicas identical after gather_table()."""
    monkeypatch.setenv("CDC_TEST_GROUPS", "cdc")
    monkeypatch.chdir(tmp_path)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "lazy", False, "sharded", True, "cdcple"), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    assert r0["losses"] == r1["losses"] and all(np.isfinite(r0["losses"]))
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), f"replicas diverged in {k}"
    from helpers import assert_close, is_pre_bn_bias
    ref_sd, ref_losses = _single_process_reference("lazy", kind="cdcple")
    for a, b in zip(r0["losses"], ref_losses):
        assert abs(a - b) < 2e-5, (r0["losses"], ref_losses)
    names = set(ref_sd)
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, names) or "num_batches" in k:
            continue
        # a running mean contains the (noise-driven, +-lr per step) bias of the Linear in front of it: 0.1 * sum over STEPS steps
        atol = 3e-3 if k.endswith("running_mean") else 2e-5
        assert_close(r0["sd"][k], v, 5e-4, atol, f"2-rank vs 1-rank: {k}")
    g = _data(world)[2]
    assert sorted(set(g.tolist())) == [0, 1, 2, 3]


def test_two_ranks_with_per_rank_batchnorm_statistics(cuda, tmp_path):
    """sync_bn=False (what bench.py runs by default for N>1: BatchNorm over the local rows, as torch DDP without
    SyncBatchNorm): the replicas still stay bit-identical in every trained tensor, and the first step's loss is the mean
    of the two single-rank losses on the shards (same initial weights, per-shard statistics)."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "lazy", True, "sharded", False), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    assert r0["losses"] == r1["losses"] and all(np.isfinite(r0["losses"]))
    for k in r0["sd"]:
        if "running_" in k:
            continue                                   # per-rank statistics: each rank tracks its own shard's
        assert torch.equal(r0["sd"][k], r1["sd"][k]), f"replicas diverged in {k}"
    sys.path.insert(0, ROOT)
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    X, y, g = _data(world)
    shard_losses = []
    for r in range(world):
        torch.manual_seed(5)
        model = MMoE(FD, 8, 3, 4, (32, 16), (8,), dropout=0.0).to(cuda).set_precision("f32")
        ts = TrainStep(model, FusedAdam(model, table_mode="lazy"), B_LOCAL)
        sl = slice(r * B_LOCAL, (r + 1) * B_LOCAL)
        bce, _ = ts.step(torch.from_numpy(X[sl]).to(cuda), torch.from_numpy(y[sl]).to(cuda), torch.from_numpy(g[sl]).to(cuda))
        shard_losses.append(float(bce.item()))
    assert abs(r0["losses"][0] - float(np.mean(shard_losses))) < 2e-6


def test_two_ranks_star_partitioned_towers(cuda, tmp_path):
    """BASELINE config C5's shape of problem: STAR (rows partitioned by domain, ragged per-domain BatchNorm groups) under data
    parallelism with the row-sharded table and global-batch statistics: replicas identical, and equal to the single-process
    step on the concatenated batch."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "lazy", False, "sharded", True, "star"), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    assert r0["losses"] == r1["losses"] and all(np.isfinite(r0["losses"]))
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), f"replicas diverged in {k}"
    from helpers import assert_close, is_pre_bn_bias
    ref_sd, ref_losses = _single_process_reference("lazy", "star")
    for a, b in zip(r0["losses"], ref_losses):
        assert abs(a - b) < 2e-5, (r0["losses"], ref_losses)
    names = set(ref_sd)
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, names) or "num_batches" in k or k == "shared_bn_bias" or (k.startswith("domain_norm.") and k.endswith(".bias")):
            continue
        atol = 5e-4 if k.endswith("running_mean") else 2e-5
        assert_close(r0["sd"][k], v, 5e-4, atol, f"star 2-rank vs 1-rank: {k}")


def test_three_ranks_row_sharded_table(cuda, tmp_path):
    """world = 3 (ownership row % 3, three sorted runs merged by the owner, list capacities that do not divide the batch):
    replicas identical and equal to the single-process step on the concatenated batch."""
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "lazy", True, "sharded"), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    for o in outs[1:]:
        assert o["losses"] == outs[0]["losses"]
        for k in outs[0]["sd"]:
            assert torch.equal(o["sd"][k], outs[0]["sd"][k]), f"replicas diverged in {k}"
    from helpers import assert_close, is_pre_bn_bias
    ref_sd, ref_losses = _single_process_reference("lazy", world=world)
    for a, b in zip(outs[0]["losses"], ref_losses):
        assert abs(a - b) < 2e-5
    names = set(ref_sd)
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, names) or "num_batches" in k:
            continue
        atol = 5e-4 if k.endswith("running_mean") else 2e-5
        assert_close(outs[0]["sd"][k], v, 5e-4, atol, f"3-rank vs 1-rank: {k}")


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_row_sharded_with_sort_and_id_exchange_one_step_ahead(cuda, tmp_path, monkeypatch, use_graph):
    """step(..., next_X=) under data parallelism: the next batch's local sort, bucketing and id exchange leave the next step's
    critical path (trainer._dp_sequence_sharded).  Same replicas, same single-process reference as without."""
    monkeypatch.setenv("CDC_TEST_AHEAD", "1")
    test_two_ranks_stay_identical(cuda, tmp_path, "lazy", use_graph, "sharded")


def test_one_rank_through_rccl_with_the_id_exchange_one_step_ahead(cuda, tmp_path, monkeypatch):
    """the same over RCCL (forced one-rank group): the look-ahead id exchange is an ASYNC all-to-all on the communicator's stream,
    awaited at the start of the next step"""
    monkeypatch.setenv("CDC_TEST_AHEAD", "1")
    monkeypatch.setenv("CDC_TEST_AHEAD_FORCE", "1")                # (a one-rank group takes part only when asked to: trainer._ahead_dp)
    test_one_rank_through_rccl(cuda, tmp_path, "sharded", True)


def test_one_rank_through_rccl_with_the_split_fused_towers(cuda, tmp_path):
    """cdc_tower_dp's six phases and the four BatchNorm all-reduces between them, issued through RCCL (forced one-rank group) and
    captured with every other launch and collective of the step in ONE graph: equal to the single-process step with the monolithic
    tower launches."""
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "lazy", True, "sharded", True, "ple64", "nccl", True), nprocs=1, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    ref_sd, ref_losses = _single_process_reference("lazy", kind="ple64", world=1)
    for a, b in zip(r0["losses"], ref_losses):
        assert abs(a - b) < 1e-5, (r0["losses"], ref_losses)
    from helpers import is_pre_bn_bias
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, set(ref_sd)) or "num_batches" in k:
            continue
        assert float((r0["sd"][k].double() - v.double()).abs().max()) < 2e-4, k


@pytest.mark.parametrize("table_dist,use_graph", [("sharded", True), ("sharded", False), ("replicated", True)])
def test_one_rank_through_rccl(cuda, tmp_path, table_dist, use_graph):
    """The collective call path itself over RCCL ("nccl" backend): a forced one-rank process group issues every all-to-all /
    all-reduce / all-gather of the data-parallel step on the communicator's stream (async row-gradient exchange); with use_graph the
    whole step — launches AND collectives — is captured and replayed as ONE graph (round 4; the worker asserts it).  With one rank
    the step must equal the single-process step."""
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "lazy", use_graph, table_dist, True, "mmoe", "nccl", True), nprocs=1, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    from helpers import assert_close, is_pre_bn_bias
    ref_sd, ref_losses = _single_process_reference("lazy", world=1)
    assert all(np.isfinite(r0["losses"]))
    for a, b in zip(r0["losses"], ref_losses):
        assert abs(a - b) < 2e-5, (r0["losses"], ref_losses)
    names = set(ref_sd)
    for k, v in ref_sd.items():
        if is_pre_bn_bias(k, names) or "num_batches" in k:
            continue
        atol = 5e-4 if k.endswith("running_mean") else 2e-5
        assert_close(r0["sd"][k], v, 5e-4, atol, f"1-rank RCCL vs single process: {k}")
