"""Kernel-level parity tests through the C-ABI (plans of one op) against plain torch fp32 references.

bf16 contractions are checked against the SAME arithmetic restated on the CPU: operands rounded to bf16
(round-to-nearest-even), products accumulated in fp32 — so the bf16 path is held to fp32-accumulation
tolerance, not to a loose "bf16 noise" bound."""
import numpy as np
import pytest
import torch

from helpers import O, assert_close

pytestmark = pytest.mark.gpu


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _plan(cuda, B, precision="f32", training=True, dropout=0.0):
    from cdcmdr_amd import plan as P
    return P, P.Plan(cuda, B, precision=precision, training=training, dropout=dropout)


# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,F,D,R", [(64, 7, 8, None), (1, 3, 16, None), (4096, 26, 16, None), (33, 5, 3, None)])
def test_gather_bit_exact(cuda, B, F, D, R):
    from cdcmdr_amd.model.layer import FeaturesEmbedding
    rng = np.random.default_rng(B + F)
    fd = [int(v) for v in rng.integers(1, 2000, size=F)]
    torch.manual_seed(1)
    emb = FeaturesEmbedding(fd, D).to(cuda)
    x = np.stack([rng.integers(0, d, size=B) for d in fd], axis=1).astype(np.int32)
    out = emb(torch.from_numpy(x).to(cuda), squeeze_dim=True)
    want = O.embed(emb.embedding_dict.weight.detach().cpu(), x, fd)
    assert torch.equal(out.cpu(), want)                                   # bit-exact row selection
    out3 = emb(torch.from_numpy(x).to(cuda))
    assert out3.shape == (B, F, D) and torch.equal(out3.cpu().flatten(1), want)


def test_gather_golden_g1(cuda):
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g1_gather.npz"))
    from cdcmdr_amd.model.layer import FeaturesEmbedding
    emb = FeaturesEmbedding(d["field_dims"].tolist(), 8).to(cuda)
    emb.load_state_dict({"embedding_dict.weight": torch.from_numpy(d["table"])})
    out = emb(torch.from_numpy(d["x"]).to(cuda), squeeze_dim=True)
    assert torch.equal(out.cpu(), torch.from_numpy(d["out"]))
    holder = next(iter(emb._cache().plans.values()))[1]
    assert np.array_equal(holder.emb_op.idx.cpu().numpy(), d["idx"])   # int32 index selection, bit-exact


@pytest.mark.parametrize("fd,B,exact", [([50, 30, 400, 20], 200, True), ([5, 3, 40, 2], 300, False)])
def test_gather_dense_grad_matches_embedding_backward(cuda, fd, B, exact):
    """Rows with fewer than 64 duplicates in the batch are summed in ascending batch order — the order of
    aten::embedding_dense_backward on the CPU — and equal torch's gradient to the last bit; hotter rows (a domain column)
    are summed in 64/D parallel parts and agree to fp32 rounding."""
    from cdcmdr_amd.model.layer import FeaturesEmbedding
    torch.manual_seed(0)
    emb = FeaturesEmbedding(fd, 4).to(cuda)
    rng = np.random.default_rng(0)
    x = np.stack([rng.integers(0, d, size=B) for d in fd], axis=1).astype(np.int32)
    if exact:
        assert max(np.bincount(x[:, f]).max() for f in range(len(fd))) < 64
    out = emb(torch.from_numpy(x).to(cuda), squeeze_dim=True)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    out.backward(g.to(cuda))
    table = emb.embedding_dict.weight.detach().cpu().clone().requires_grad_(True)
    O.embed(table, x, fd).backward(g)
    if exact:
        assert torch.equal(emb.embedding_dict.weight.grad.cpu(), table.grad)
    else:
        assert_close(emb.embedding_dict.weight.grad, table.grad, 1e-5, 1e-5, "dense table gradient, hot rows")


def test_gather_out_of_range_sets_flag(cuda):
    from cdcmdr_amd.model.layer import FeaturesEmbedding
    emb = FeaturesEmbedding([4, 4], 4).to(cuda)
    x = torch.tensor([[0, 1], [3, 9]], dtype=torch.int32, device=cuda)   # 9 + 4 = 13 >= 8 rows
    out = emb(x, squeeze_dim=True)
    holder = next(iter(emb._cache().plans.values()))[1]
    assert int(holder.emb_op.err.item()) == 4                             # 1 + flat position of the bad id
    assert float(out[1, 4:].abs().max()) == 0.0


# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,Ns", [(64, 24, [32, 16, 4]), (257, 416, [256, 8]), (1, 40, [7]), (130, 65, [129, 3, 64])])
def test_glinear_fwd_bwd(cuda, precision, M, K, Ns):
    P, plan = _plan(cuda, M, precision)
    gen = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=gen)
    xb = plan.new(K)
    xb.tensor().copy_(x)
    ws = [torch.nn.Parameter((torch.randn(n, K, generator=gen) / K ** 0.5).to(cuda)) for n in Ns]
    bs = [torch.nn.Parameter(torch.randn(n, generator=gen).to(cuda)) for n in Ns]
    # first group: relu; all groups share x -> grad-input reduces over the groups inside one kernel
    op = P.GLinear(plan, [{"x": xb, "w": w, "b": b, "act_cols": (w.shape[0] if i == 0 else 0)} for i, (w, b) in enumerate(zip(ws, bs))],
                   relu=True)
    plan.finalize(op.outs)
    plan.forward()
    gouts = [torch.randn(M, n, generator=gen) for n in Ns]
    for o, g in zip(op.outs, gouts):
        o.grad.tensor().copy_(g)
    # the gradient buffer of a relu-fused output holds dZ: apply the mask the consumer kernel would have applied
    y0 = op.outs[0].tensor().cpu()
    op.outs[0].grad.tensor().mul_((y0 > 0).float().to(cuda))
    plan.backward()
    rnd = _bf16_round if precision == "bf16" else (lambda t: t)
    xr = x.clone().requires_grad_(True)
    tot = 0
    outs_ref = []
    wr = [w.detach().cpu().clone().requires_grad_(True) for w in ws]
    br = [b.detach().cpu().clone().requires_grad_(True) for b in bs]
    for i in range(len(Ns)):
        y = rnd(xr) @ rnd(wr[i]).t() + br[i]
        y = torch.relu(y) if i == 0 else y
        outs_ref.append(y)
        tot = tot + (y * gouts[i]).sum()
    tot.backward()
    for i, o in enumerate(op.outs):
        assert_close(o.tensor(), outs_ref[i], 2e-5, 2e-5, f"y{i}")
    if precision == "f32":
        assert_close(xb.grad.tensor(), xr.grad, 1e-4, 1e-4, "dx")
        for i in range(len(Ns)):
            assert_close(plan.param_grads[id(ws[i])], wr[i].grad, 1e-4, 1e-4, f"dw{i}")
            assert_close(plan.param_grads[id(bs[i])], br[i].grad, 1e-4, 1e-4, f"db{i}")
    else:
        # backward contractions round THEIR operands (dZ, W, X) to bf16: restate them the same way
        dzs = [gouts[0] * (outs_ref[0] > 0).float()] + gouts[1:]
        dx = sum(_bf16_round(dz) @ _bf16_round(w.detach().cpu()) for dz, w in zip(dzs, ws))
        assert_close(xb.grad.tensor(), dx, 1e-4, 1e-4, "dx")
        for i in range(len(Ns)):
            assert_close(plan.param_grads[id(ws[i])], _bf16_round(dzs[i]).t() @ _bf16_round(x), 1e-4, 2e-4, f"dw{i}")
            # bias gradient: the column sums of the SAME bf16 dZ the weight gradient contracts (one MFMA against a fragment of ones)
            assert_close(plan.param_grads[id(bs[i])], _bf16_round(dzs[i]).sum(0), 1e-5, 1e-4, f"db{i}")


def test_glinear_dropout_statistics_and_backward_mask(cuda):
    P, plan = _plan(cuda, 512, "f32", training=True, dropout=0.2)
    x = plan.new(64)
    x.tensor().copy_(torch.rand(512, 64) + 0.5)                          # positive inputs, positive weights -> relu never clamps
    w = torch.nn.Parameter(torch.rand(128, 64, device=cuda) / 64)
    op = P.GLinear(plan, [{"x": x, "w": w, "b": None}], relu=True, dropout=True)
    plan.finalize(op.outs)
    plan.forward()
    y = op.outs[0].tensor().cpu()
    ref = x.tensor().cpu() @ w.detach().cpu().t()
    kept = y != 0
    keep_rate = kept.float().mean().item()
    assert abs(keep_rate - 0.8) < 0.01                                   # golden g8: torch's dropout keeps 1-p
    assert_close(y[kept], (ref / 0.8)[kept], 1e-4, 1e-5, "kept values are scaled by 1/(1-p)")
    plan.step_dev.add_(1)                                                # next step -> a different mask
    plan.forward()
    y2 = op.outs[0].tensor().cpu()
    assert ((y2 != 0) != kept).float().mean().item() > 0.2


# --------------------------------------------------------------------------------------------------
def test_gate_pool_fwd_bwd(cuda):
    B, E, H = 130, 5, 24
    P, plan = _plan(cuda, B)
    gen = torch.Generator().manual_seed(3)
    ex = plan.new(E * H)
    exv = torch.randn(B, E * H, generator=gen)
    ex.tensor().copy_(exv)
    sels = [[0, 1, 4], [2, 3, 4], [0, 1, 2, 3, 4]]
    lgs = []
    for s in sels:
        lb = plan.new(len(s))
        lb.tensor().copy_(torch.randn(B, len(s), generator=gen))
        lgs.append(lb)
    op = P.GatePool(plan, ex, E, H, list(zip(lgs, sels)))
    plan.finalize(op.outs)
    plan.forward()
    gouts = [torch.randn(B, H, generator=gen) for _ in sels]
    for o, g in zip(op.outs, gouts):
        o.grad.tensor().copy_(g)
    plan.backward()
    exr = exv.clone().requires_grad_(True)
    lr = [l.tensor().cpu().clone().requires_grad_(True) for l in lgs]
    tot = 0
    for i, s in enumerate(sels):
        p = torch.softmax(lr[i], dim=1)
        cat = exr.view(B, E, H)[:, s, :]
        out = (p.unsqueeze(-1) * cat).sum(1)
        assert_close(op.outs[i].tensor(), out, 1e-5, 1e-6, f"pool out {i}")
        tot = tot + (out * gouts[i]).sum()
    tot.backward()
    assert_close(ex.grad.tensor(), exr.grad, 1e-5, 1e-6, "d experts")
    for i in range(len(sels)):
        assert_close(lgs[i].grad.tensor(), lr[i].grad, 1e-4, 1e-6, f"d logits {i}")


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("M", [1, 2, 64, 200])
def test_batchnorm_fwd_bwd(cuda, training, M):
    P, plan = _plan(cuda, M, training=training)
    gen = torch.Generator().manual_seed(M)
    Cs = [70, 8]
    bns = [torch.nn.BatchNorm1d(c).to(cuda) for c in Cs]
    for bn in bns:
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    xs = [torch.randn(M, c, generator=gen) * 2 + 1 for c in Cs]
    segs = []
    for xv, bn in zip(xs, bns):
        xb = plan.new(xv.shape[1])
        xb.tensor().copy_(xv)
        segs.append({"x": xb, "gamma": bn.weight, "beta": bn.bias, "gamma_param": bn.weight, "beta_param": bn.bias,
                     "running_mean": bn.running_mean, "running_var": bn.running_var,
                     "num_batches_tracked": bn.num_batches_tracked})
    refs = [torch.nn.BatchNorm1d(c) for c in Cs]
    for r, bn in zip(refs, bns):
        r.load_state_dict({k: v.cpu() for k, v in bn.state_dict().items()})
        r.train(training)
    op = P.BatchNorm(plan, segs, relu=True, dropout=False)
    plan.finalize(op.outs)
    plan.forward()
    gouts = [torch.randn(M, c, generator=gen) for c in Cs]
    for o, g in zip(op.outs, gouts):
        o.grad.tensor().copy_(g)
    plan.backward()
    for i, (r, xv) in enumerate(zip(refs, xs)):
        xr = xv.clone().requires_grad_(True)
        y = torch.relu(xr if M == 1 else r(xr))                 # the reference skips BatchNorm for one row
        (y * gouts[i]).sum().backward()
        assert_close(op.outs[i].tensor(), y, 1e-4, 1e-5, f"bn out {i}")
        assert_close(segs[i]["x"].grad.tensor(), xr.grad, 2e-4, 2e-5, f"bn dx {i}")
        if M > 1:
            assert_close(plan.param_grads[id(bns[i].weight)], r.weight.grad, 2e-4, 2e-5, f"dgamma {i}")
            assert_close(plan.param_grads[id(bns[i].bias)], r.bias.grad, 2e-4, 2e-5, f"dbeta {i}")
            assert_close(bns[i].running_mean, r.running_mean, 1e-5, 1e-6, "running_mean")
            assert_close(bns[i].running_var, r.running_var, 1e-5, 1e-6, "running_var")
            assert int(bns[i].num_batches_tracked) == int(r.num_batches_tracked)
        else:
            assert int(bns[i].num_batches_tracked) == 0


def test_rowdot_towers_with_addend(cuda):
    B, K = 100, 33
    P, plan = _plan(cuda, B)
    gen = torch.Generator().manual_seed(9)
    xs = [torch.randn(B, K, generator=gen) for _ in range(3)]
    bufs = []
    for xv in xs:
        b = plan.new(K)
        b.tensor().copy_(xv)
        bufs.append(b)
    wide_in = plan.new(K)
    wide_in.tensor().copy_(torch.randn(B, K, generator=gen))
    lin_w = torch.nn.Linear(K, 1).to(cuda)
    wide = P.RowDot(plan, [{"x": wide_in, "w": lin_w.weight, "b": lin_w.bias}])
    lins = [torch.nn.Linear(K, 1).to(cuda) for _ in range(3)]
    out = plan.new(3)
    op = P.RowDot(plan, [{"x": b, "w": l.weight, "b": l.bias, "out": out.slice(i, i + 1)} for i, (b, l) in enumerate(zip(bufs, lins))],
                  addends=[wide.outs[0]], sigmoid=True)
    plan.finalize([out])
    plan.forward()
    g = torch.randn(B, 3, generator=gen)
    out.grad.tensor().copy_(g)
    plan.backward()
    xr = [xv.clone().requires_grad_(True) for xv in xs]
    wr = wide_in.tensor().cpu().clone().requires_grad_(True)
    lw = torch.nn.Linear(K, 1)
    lw.load_state_dict({k: v.cpu() for k, v in lin_w.state_dict().items()})
    refl = []
    for l in lins:
        r = torch.nn.Linear(K, 1)
        r.load_state_dict({k: v.cpu() for k, v in l.state_dict().items()})
        refl.append(r)
    wl = lw(wr)
    y = torch.cat([torch.sigmoid(refl[i](xr[i]) + wl) for i in range(3)], dim=1)
    (y * g).sum().backward()
    assert_close(out.tensor(), y, 1e-5, 1e-6, "towers out")
    for i in range(3):
        assert_close(bufs[i].grad.tensor(), xr[i].grad, 1e-4, 1e-6, f"dx{i}")
        assert_close(plan.param_grads[id(lins[i].weight)], refl[i].weight.grad, 1e-4, 1e-5, f"dw{i}")
        assert_close(plan.param_grads[id(lins[i].bias)], refl[i].bias.grad, 1e-4, 1e-5, f"db{i}")
    assert_close(wide_in.grad.tensor(), wr.grad, 1e-4, 1e-6, "d wide in")
    assert_close(plan.param_grads[id(lin_w.weight)], lw.weight.grad, 1e-4, 1e-5, "d wide w")
    assert_close(plan.param_grads[id(lin_w.bias)], lw.bias.grad, 1e-4, 1e-5, "d wide b")


def test_bce_golden_g7(cuda):
    import ctypes as C
    import os
    from cdcmdr_amd import _lib as L
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_bce.npz"))
    lib = L.load()
    p = torch.from_numpy(d["p"]).to(cuda).reshape(-1, 1).contiguous()
    y = torch.from_numpy(d["y"]).to(cuda).contiguous()
    n = p.shape[0]
    loss = torch.zeros(1, device=cuda)
    dp = torch.empty_like(p)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.cdc_bce_fwd_bwd(p.data_ptr(), 1, None, None, y.data_ptr(), loss.data_ptr(), dp.data_ptr(), 1, n, 1, 1.0 / n, s), "bce")
    assert_close(loss, d["loss"].reshape(1), 1e-6, 1e-6, "bce loss incl. the -100 clamp")
    assert_close(dp.reshape(-1), d["dp"], 1e-5, 0.0, "bce gradient incl. the 1e-12 clamp")


@pytest.mark.parametrize("with_scratch", [False, True])
@pytest.mark.parametrize("B", [1, 100, 1024, 1025, 4096, 5000, 16384, 16385, 20000, 32768])
def test_sort_dedupe_against_numpy(cuda, B, with_scratch):
    """Per-field sort + dedupe — single-workgroup LDS path (no scratch, up to 16384 rows) and chunked path (sorted runs
    merged by rank; any size up to 32768 = the gathered batch of an 8-GPU step) — integer work, bit-exact against numpy's
    stable sort."""
    if not with_scratch and B > 16384:
        pytest.skip("needs the scratch buffer")
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    F = 5
    rng = np.random.default_rng(B)
    vocab = [3, 50, 1000, 100000, 7]
    idx = np.stack([rng.integers(0, v, size=B) + 1000 * f for f, v in enumerate(vocab)], axis=1).astype(np.int32)
    d_idx = torch.from_numpy(idx).to(cuda)
    uniq = torch.full((F, B), -1, dtype=torch.int32, device=cuda)
    seg = torch.full((F, B + 1), -1, dtype=torch.int32, device=cuda)
    perm = torch.full((F, B), -1, dtype=torch.int32, device=cuda)
    cnt = torch.zeros(F, dtype=torch.int32, device=cuda)
    scratch = torch.empty(2 * F * B, dtype=torch.int64, device=cuda) if with_scratch else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.cdc_embed_sort_dedupe(d_idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                      None if scratch is None else scratch.data_ptr(), B, F, s), "sort")
    uniq, seg, perm, cnt = uniq.cpu().numpy(), seg.cpu().numpy(), perm.cpu().numpy(), cnt.cpu().numpy()
    for f in range(F):
        order = np.argsort(idx[:, f], kind="stable")                     # ascending row, ascending batch position inside
        assert np.array_equal(perm[f], order)
        rows, starts = np.unique(idx[order, f], return_index=True)
        n = len(rows)
        assert cnt[f] == n and np.array_equal(uniq[f, :n], rows)
        assert np.array_equal(seg[f, :n], starts) and seg[f, n] == B


@pytest.mark.parametrize("B,N,cap", [(64, 2, 64), (4096, 8, 832), (1000, 3, 400), (512, 4, 20)])
def test_shard_bucket_expand_pack(cuda, B, N, cap):
    """Row-sharded table exchange helpers (include/cdcmdr.h cdc_shard_*): bucket the unique rows of a batch by owner
    (row % N) into [N][cap][F] lists, expand the rows the owners return into the gathered batch, pack per-row gradients
    for the way back.  Integer/copy work: bit-exact against numpy; ids < 0 (out of range upstream) are skipped."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    F, D = 5, 8
    rng = np.random.default_rng(B + N)
    vocab = [3, 50, 1000, 100000, 7]
    idx = np.stack([rng.integers(0, v, size=B) + 1000 * f for f, v in enumerate(vocab)], axis=1).astype(np.int32)
    idx[rng.integers(0, B, size=3), 2] = -1                              # what cdc_embed_index writes for a bad id
    d_idx = torch.from_numpy(idx).to(cuda)
    uniq = torch.full((F, B), -7, dtype=torch.int32, device=cuda)
    seg = torch.zeros((F, B + 1), dtype=torch.int32, device=cuda)
    perm = torch.zeros((F, B), dtype=torch.int32, device=cuda)
    cnt = torch.zeros(F, dtype=torch.int32, device=cuda)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.cdc_embed_sort_dedupe(d_idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(), None, B, F, s), "sort")
    send = torch.full((N, cap, F), -5, dtype=torch.int32, device=cuda)
    slot_of = torch.full((F, B), -9, dtype=torch.int32, device=cuda)
    over = torch.zeros(1, dtype=torch.int32, device=cuda)
    L.check(lib.cdc_shard_bucket(uniq.data_ptr(), cnt.data_ptr(), send.data_ptr(), slot_of.data_ptr(), over.data_ptr(), B, F, N, cap, s), "bucket")
    send_h, slot_h, over_h = send.cpu().numpy(), slot_of.cpu().numpy(), int(over.item())
    want_over = 0
    for f in range(F):
        rows = np.unique(idx[:, f])
        rows = rows[rows >= 0]
        for o in range(N):
            mine = rows[rows % N == o]                                   # ascending
            kept = mine[:cap]
            if len(mine) > cap:
                want_over = max(want_over, len(mine))
            got = send_h[o, :, f]
            assert np.array_equal(got[:len(kept)], kept) and np.all(got[len(kept):] == -1)
    assert over_h == want_over
    if want_over:
        return
    # owners answer with a function of the row id; the expansion must equal a plain gather of that "table"
    def row_vec(rows):
        return (rows[..., None].astype(np.float32) * 0.5 + np.arange(D, dtype=np.float32)) * (rows[..., None] >= 0)
    rows_recv = torch.from_numpy(row_vec(send_h.astype(np.int64))).to(cuda).contiguous()        # [N, cap, F, D]
    out = torch.full((B, F * D), 123.0, dtype=torch.float32, device=cuda)
    L.check(lib.cdc_shard_expand(rows_recv.data_ptr(), uniq.data_ptr(), cnt.data_ptr(), seg.data_ptr(), perm.data_ptr(),
                                 slot_of.data_ptr(), out.data_ptr(), B, F, D, N, cap, s), "expand")
    assert np.array_equal(out.cpu().numpy().reshape(B, F, D), row_vec(idx.astype(np.int64)))
    # way back: per-unique-row gradients land in the slot their row id was sent in
    rowgrad = torch.randn((F, B, D), dtype=torch.float32, device=cuda)
    gsend = torch.zeros((N, cap, F, D), dtype=torch.float32, device=cuda)
    L.check(lib.cdc_shard_pack(rowgrad.data_ptr(), uniq.data_ptr(), cnt.data_ptr(), slot_of.data_ptr(), gsend.data_ptr(), B, F, D, N, cap, s), "pack")
    g_h, rg_h, uniq_h, cnt_h = gsend.cpu().numpy(), rowgrad.cpu().numpy(), uniq.cpu().numpy(), cnt.cpu().numpy()
    want = np.zeros_like(g_h)
    for f in range(F):
        for j in range(cnt_h[f]):
            r = uniq_h[f, j]
            if r >= 0:
                want[r % N, slot_h[f, j], f] = rg_h[f, j]
    assert np.array_equal(g_h, want)


@pytest.mark.parametrize("n_runs,run_len", [(2, 64), (8, 832), (3, 100), (1, 50)])
def test_merge_dedupe_equals_sort_dedupe(cuda, n_runs, run_len):
    """cdc_embed_merge_dedupe (rank merge of already sorted runs, used by the owner of a row shard) gives exactly what the
    general sort gives on the same batch, -1 padding included."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    F, B = 4, n_runs * run_len
    rng = np.random.default_rng(n_runs * 1000 + run_len)
    idx = np.full((n_runs, run_len, F), -1, dtype=np.int32)
    for r in range(n_runs):
        for f, v in enumerate([5, 300, 100000, 40]):
            n = int(rng.integers(0, run_len + 1))
            idx[r, :n, f] = np.sort(rng.choice(max(v, n), size=n, replace=False) if v >= n else rng.integers(0, v, size=n))
            if v < n:                       # duplicates inside a run cannot come from the bucket kernel; keep them unique
                u = np.unique(idx[r, :n, f])
                idx[r, :, f] = -1
                idx[r, :len(u), f] = u
    d_idx = torch.from_numpy(idx.reshape(B, F)).to(cuda)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for merge in (False, True):
        uniq = torch.full((F, B), -7, dtype=torch.int32, device=cuda)
        seg = torch.full((F, B + 1), -7, dtype=torch.int32, device=cuda)
        perm = torch.full((F, B), -7, dtype=torch.int32, device=cuda)
        cnt = torch.zeros(F, dtype=torch.int32, device=cuda)
        scratch = torch.empty(2 * F * B, dtype=torch.int64, device=cuda)
        if merge:
            L.check(lib.cdc_embed_merge_dedupe(d_idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                               scratch.data_ptr(), B, F, n_runs, s), "merge")
        else:
            L.check(lib.cdc_embed_sort_dedupe(d_idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                              scratch.data_ptr(), B, F, s), "sort")
        c = cnt.cpu().numpy()
        outs.append((c, [uniq[f, :c[f]].cpu().numpy() for f in range(F)], [seg[f, :c[f] + 1].cpu().numpy() for f in range(F)],
                     perm.cpu().numpy()))
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
    for f in range(F):
        assert np.array_equal(a[1][f], b[1][f]) and np.array_equal(a[2][f], b[2][f])
    # the owner's per-row sums without the sorted copy == the general segment sum (bit for bit) on every real row
    D = 16
    g = torch.from_numpy(rng.standard_normal((B, F * D)).astype(np.float32)).to(cuda)
    sorted_scratch = torch.empty((F, B, D), dtype=torch.float32, device=cuda)
    rg_a = torch.zeros((F, B, D), dtype=torch.float32, device=cuda)
    rg_b = torch.full((F, B, D), 7.0, dtype=torch.float32, device=cuda)
    L.check(lib.cdc_embed_segment_sum(g.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(), sorted_scratch.data_ptr(),
                                      rg_a.data_ptr(), B, F, D, s), "segment_sum")
    L.check(lib.cdc_embed_segment_sum_direct(g.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(), uniq.data_ptr(),
                                             rg_b.data_ptr(), B, F, D, s), "segment_sum_direct")
    c = cnt.cpu().numpy()
    for f in range(F):
        real = int((uniq[f, :c[f]] >= 0).sum())
        assert torch.equal(rg_a[f, :real], rg_b[f, :real])
        assert bool((rg_b[f, real:] == 7.0).all())                 # padding segment and unused slots stay untouched


@pytest.mark.parametrize("B", [300, 4096])
def test_sort_dedupe_from_raw_ids_with_begin_step(cuda, B):
    """cdc_embed_sort_dedupe_ids == cdc_embed_index + cdc_embed_sort_dedupe (out-of-range ids sort as -1), and its first
    launch does cdc_begin_step's work."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    F = 5
    fd = np.array([7, 1000, 3, 50000, 90], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(fd)[:-1]]).astype(np.int32)
    R = int(fd.sum())
    rng = np.random.default_rng(B)
    ids = np.stack([rng.integers(0, d, size=B) for d in fd], axis=1).astype(np.int32)
    ids[3, 4] = 10 ** 6                                   # past the table
    ids[5, 0] = -9                                        # before it
    d_ids, d_off = torch.from_numpy(ids).to(cuda), torch.from_numpy(offsets).to(cuda)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    step = torch.tensor([41], dtype=torch.int32, device=cuda)
    acc = torch.full((2,), 3.5, dtype=torch.float64, device=cuda)
    for fused in (False, True):
        uniq = torch.full((F, B), -7, dtype=torch.int32, device=cuda)
        seg = torch.full((F, B + 1), -7, dtype=torch.int32, device=cuda)
        perm = torch.full((F, B), -7, dtype=torch.int32, device=cuda)
        cnt = torch.zeros(F, dtype=torch.int32, device=cuda)
        scratch = torch.empty(2 * F * B, dtype=torch.int64, device=cuda)
        if fused:
            err = torch.zeros(1, dtype=torch.int32, device=cuda)
            L.check(lib.cdc_embed_sort_dedupe_ids(d_ids.data_ptr(), d_off.data_ptr(), R, step.data_ptr(), acc.data_ptr(), 2,
                                                  err.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                                  scratch.data_ptr(), B, F, s), "sort ids")
            assert int(err.item()) == 5 * F + 0 + 1                # the later of the two bad positions (atomicMax), 1-based
        else:
            idx = torch.empty((B, F), dtype=torch.int32, device=cuda)
            L.check(lib.cdc_embed_index(d_ids.data_ptr(), d_off.data_ptr(), idx.data_ptr(), None, B, F, R, s), "index")
            L.check(lib.cdc_embed_sort_dedupe(idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                              scratch.data_ptr(), B, F, s), "sort")
        c = cnt.cpu().numpy()
        outs.append((c, [uniq[f, :c[f]].cpu().numpy() for f in range(F)], [seg[f, :c[f] + 1].cpu().numpy() for f in range(F)],
                     perm.cpu().numpy()))
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
    for f in range(F):
        assert np.array_equal(a[1][f], b[1][f]) and np.array_equal(a[2][f], b[2][f])
    assert a[1][4][-1] == -1 and a[1][0][-1] == -1        # the two bad ids
    assert int(step.item()) == 42 and float(acc.abs().sum()) == 0.0


def test_add_n_matches_sequential_adds(cuda):
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    rows, n = 1000, 3
    torch.manual_seed(3)
    srcs = [torch.randn(rows, 5, device=cuda) for _ in range(n)]           # column 2 of a wider buffer each
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for accumulate in (0, 1):
        dst = torch.randn(rows, 4, device=cuda)
        ref = dst.clone()
        if accumulate:
            for t in srcs:
                ref[:, 1] += t[:, 2]
        else:
            ref[:, 1] = srcs[0][:, 2]
            for t in srcs[1:]:
                ref[:, 1] += t[:, 2]
        a = L.AddNArgs()
        a.dst, a.ld_dst, a.rows, a.cols, a.n, a.accumulate = dst.data_ptr() + 4, 4, rows, 1, n, accumulate
        for k, t in enumerate(srcs):
            a.src[k], a.ld_src[k] = t.data_ptr() + 8, 5
        L.check(lib.cdc_add_n(C.byref(a), s), "add_n")
        assert torch.equal(dst, ref)


@pytest.mark.parametrize("D", [3, 4, 16, 32, 64, 96, 128])
def test_segment_sum_every_length_class(cuda, D):
    """cdc_embed_segment_sum over segments of every class the launch distinguishes (thread per chunk below 8 entries, a wave per
    row from 8 on — split into sub-lane parts from 64 on when D divides 64 — the whole workgroup from 512 on): segments
    shorter than 64 equal the sequential fp32 sum in ascending batch order to the last bit, longer ones to fp32 rounding."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    lens = [1, 2, 7, 8, 9, 63, 64, 65, 200, 511, 512, 513, 1500]
    F = 2
    rows0 = np.repeat(np.arange(len(lens)), lens)                    # field 0: one row per length class
    B = len(rows0)
    rng = np.random.default_rng(D)
    perm0 = rng.permutation(B)
    idx = np.empty((B, F), dtype=np.int32)
    idx[:, 0] = rows0[perm0] * 3 + 5
    idx[:, 1] = rng.integers(0, 50, size=B) + 10_000                 # field 1: many medium segments
    g = rng.standard_normal((B, F * D)).astype(np.float32)
    d_idx, d_g = torch.from_numpy(idx).to(cuda), torch.from_numpy(g).to(cuda)
    uniq = torch.empty((F, B), dtype=torch.int32, device=cuda)
    seg = torch.empty((F, B + 1), dtype=torch.int32, device=cuda)
    perm = torch.empty((F, B), dtype=torch.int32, device=cuda)
    cnt = torch.zeros(F, dtype=torch.int32, device=cuda)
    scratch = torch.empty(2 * F * B, dtype=torch.int64, device=cuda)
    rg = torch.full((F, B, D), 7.0, dtype=torch.float32, device=cuda)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.cdc_embed_sort_dedupe(d_idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                      scratch.data_ptr(), B, F, s), "sort")
    L.check(lib.cdc_embed_segment_sum(d_g.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(), None, rg.data_ptr(), B, F, D, s),
            "segment_sum")
    c = cnt.cpu().numpy()
    u, out = uniq.cpu().numpy(), rg.cpu().numpy()
    for f in range(F):
        assert c[f] == len(np.unique(idx[:, f]))
        for j in range(c[f]):
            rows = np.nonzero(idx[:, f] == u[f, j])[0]               # ascending batch order
            vals = g[rows, f * D:(f + 1) * D]
            if len(rows) < 64:
                acc = np.zeros(D, dtype=np.float32)
                for v in vals:
                    acc = (acc + v).astype(np.float32)
                assert np.array_equal(out[f, j], acc), (f, j, len(rows))
            else:
                ref = vals.astype(np.float64).sum(0)
                assert np.allclose(out[f, j], ref, rtol=2e-5, atol=2e-5 * np.sqrt(len(rows))), (f, j, len(rows))
        assert (out[f, c[f]:] == 7.0).all()                          # slots past the unique rows stay untouched


def test_adam_multi_table_equals_the_argument_block_launches_and_torch_adam(cuda):
    """cdc_adam_multi (<= 48 tensors per launch, descriptors as kernel arguments) and cdc_adam_multi_table (any number, descriptors
    and workgroup map in device memory) on the same 70 tensors — whole chunks, ragged tails, unaligned sizes, a gradient given as
    three split-K slabs — bit-identical, and equal to torch.optim.Adam with the L2 term in the gradient (run.py:720-721)."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    lib = L.load()
    g_ = torch.Generator().manual_seed(3)
    sizes = [5000, 4096, 8192 + 7, 33, 1, 4097, 12288] * 10
    lr, b1, b2, eps, wd, l2 = 1e-3, 0.9, 0.99, 1e-8, 1e-8, 1e-5
    w0 = [torch.randn(n, generator=g_) for n in sizes]
    gr = [torch.randn(n, generator=g_) * 0.1 for n in sizes]
    m0 = [torch.randn(n, generator=g_) * 0.01 for n in sizes]
    v0 = [torch.rand(n, generator=g_) * 1e-3 for n in sizes]
    t = 7                                                             # the step being taken
    from cdcmdr_amd.optim import step_scalar_table
    scal = step_scalar_table(lr, b1, b2, n=32).to(cuda).contiguous()         # row t: lr / (1 - b1^t), sqrt(1 - b2^t)
    step_dev = torch.full((1,), t, dtype=torch.int32, device=cuda)
    # a gradient as three slabs whose in-order sum is the gradient (tensor 2)
    parts = [gr[2] * 0.5, gr[2] * 0.25, gr[2] - gr[2] * 0.5 - gr[2] * 0.25]
    slab_stride = sizes[2] + 9
    slabs = torch.zeros(3 * slab_stride)
    for s_, p_ in enumerate(parts):
        slabs[s_ * slab_stride:s_ * slab_stride + sizes[2]] = p_
    g2_sum = (torch.zeros_like(parts[0]) + parts[0] + parts[1]) + parts[2]
    slabs = slabs.to(cuda)

    def run(table):
        w = [x.clone().to(cuda) for x in w0]
        m = [x.clone().to(cuda) for x in m0]
        v = [x.clone().to(cuda) for x in v0]
        g = [x.clone().to(cuda) for x in gr]
        reg = torch.zeros(1, dtype=torch.float64, device=cuda)

        def fill(T, i):
            T.w, T.g, T.m, T.v, T.n, T.l2 = w[i].data_ptr(), g[i].data_ptr(), m[i].data_ptr(), v[i].data_ptr(), sizes[i], l2
            if i == 2:
                T.slabs, T.slab_stride, T.n_slabs = slabs.data_ptr(), slab_stride, 3
            else:
                T.slabs, T.slab_stride, T.n_slabs = None, 0, 0

        def header(a):
            a.lerp_w, a.beta2, a.one_minus_beta2, a.eps, a.weight_decay = 1 - b1, b2, 1 - b2, eps, wd
            a.step_scalars, a.n_scalars, a.grad_scale, a.step_dev, a.reg_sum, a.reg_seed = scal.data_ptr(), 32, 1.0, step_dev.data_ptr(), reg.data_ptr(), None
        st = torch.cuda.current_stream().cuda_stream
        if table:
            tab = (L.AdamTensor * len(sizes))()
            wg_t, wg_c = [], []
            for i in range(len(sizes)):
                fill(tab[i], i)
                nck = -(-sizes[i] // L.ADAM_CHUNK)
                wg_t += [i] * nck
                wg_c += list(range(nck))
            hdr = L.AdamArgs()
            header(hdr)
            dev_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(cuda)
            wt, wc = torch.tensor(wg_t, dtype=torch.int32, device=cuda), torch.tensor(wg_c, dtype=torch.int32, device=cuda)
            assert lib.cdc_adam_multi_table(C.byref(hdr), dev_tab.data_ptr(), wt.data_ptr(), wc.data_ptr(), len(wg_t), st) == 0
        else:
            for c0 in range(0, len(sizes), L.MAX_TENSORS):
                a = L.AdamArgs()
                header(a)
                n = min(L.MAX_TENSORS, len(sizes) - c0)
                a.n_tensors = n
                for i in range(n):
                    fill(a.t[i], c0 + i)
                assert lib.cdc_adam_multi(C.byref(a), st) == 0
        torch.cuda.synchronize()
        return [x.cpu() for x in w], [x.cpu() for x in m], [x.cpu() for x in v], float(reg.item())
    wa, ma, va, ra = run(False)
    wb, mb, vb, rb = run(True)
    for i in range(len(sizes)):
        assert torch.equal(wa[i], wb[i]) and torch.equal(ma[i], mb[i]) and torch.equal(va[i], vb[i]), f"tensor {i} ({sizes[i]} elements)"
    assert abs(ra - rb) <= 1e-12 * abs(ra)
    # torch.optim.Adam on w with grad + 2*l2*w (weight_decay adds wd*w itself), from the same moments at step t
    for i in (0, 2, 3, 5):
        p = torch.nn.Parameter(w0[i].clone())
        opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
        opt.state[p] = {"step": torch.tensor(float(t - 1)), "exp_avg": m0[i].clone(), "exp_avg_sq": v0[i].clone()}
        gi = g2_sum if i == 2 else gr[i]
        p.grad = gi + 2 * l2 * w0[i]
        opt.step()
        assert_close(wb[i], p.detach(), 2e-6, 2e-7, f"torch.optim.Adam, tensor {i}")
    # (w*w is formed in fp32 and l2 is a float: 1e-7 relative)
    assert abs(rb - l2 * sum(float((x.double() ** 2).sum()) for x in w0)) <= 1e-6 * rb
