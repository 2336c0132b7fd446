"""The training step at BASELINE.json's full size (configs[1]: PLE 3-domain, 26 fields x vocab 1 M, emb_dim 16, batch 4096:
a 26 M-row table) checked through size-independent properties — the oracle needs ~3.4 s per step at this size, so it
only checks the first step's loss and one dense layer here:

  * run-to-run determinism: the same three steps twice give bit-identical parameters, table rows and losses (no float
    atomics anywhere in the step);
  * the two table optimisers agree: lazy (exact replay) == dense, bit for bit, on every row the batches touched and on a
    sample of rows they did not (the F3 semantics: untouched rows still move every step);
  * hipGraph replay == eager;
  * the gather is an exact row copy and the per-row gradient of the table equals an index_add of the batch gradient.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import O

pytestmark = pytest.mark.gpu
FIELDS, VOCAB, D, B = 26, 1_000_000, 16, 4096


def _model(cuda, precision="f32", dropout=0.0):
    from cdcmdr_amd.model.ple import PLE
    torch.manual_seed(2000)
    with torch.device(cuda):
        m = PLE([VOCAB] * FIELDS, D, 3, 2, 2, ((256, 128), (64,)), (64, 32), dropout=dropout)
    return m.set_precision(precision)


def _batches(n, seed=1):
    from cdcmdr_amd.synth import make_dataset
    X, y = make_dataset(B * n, [VOCAB] * FIELDS, n_domain=3, domain_idx=10, seed=seed)
    return X.reshape(n, B, FIELDS), y.reshape(n, B), X[:, 10].astype(np.int64).reshape(n, B)


def _run(cuda, table_mode, use_graph=False, n=4, fast_replay=False, precision="f32", dropout=0.0, announce=False):
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.trainer import TrainStep
    model = _model(cuda, precision, dropout)
    opt = FusedAdam(model, table_mode=table_mode, fast_replay=fast_replay, flush_every=2)
    ts = TrainStep(model, opt, B, use_graph=use_graph)
    X, y, g = _batches(n)
    losses = []
    Xd = [torch.from_numpy(X[s]).to(cuda) for s in range(n)]
    for s in range(n):
        # announce: every step names the next batch, whose rows are then sorted on the side chain of this step (trainer._sort_ahead)
        nxt = Xd[s + 1] if (announce and s + 1 < n) else None
        bce, _ = ts.step(Xd[s], torch.from_numpy(y[s]).to(cuda), torch.from_numpy(g[s]).to(cuda), next_X=nxt)
        losses.append(bce.clone())
    if announce:
        assert getattr(ts, "_ahead_ok", False) and ts._parity == (n - 1) % 2, "the look-ahead sort was not used"
    ts.check_ids()
    opt.flush_table()
    touched = np.unique((X + np.arange(FIELDS, dtype=np.int64) * VOCAB).reshape(-1))
    rng = np.random.default_rng(0)
    sample = np.unique(np.concatenate([touched[rng.integers(0, len(touched), 20000)], rng.integers(0, FIELDS * VOCAB, 20000)]))
    idx = torch.from_numpy(sample).to(cuda)
    table = model.embedding.embedding_dict.weight.detach()
    out = {"rows": table[idx].cpu(), "m": opt.table_m[idx].cpu(), "v": opt.table_v[idx].cpu(),
           "losses": torch.stack(losses).cpu(),
           "dense": {k: v.detach().cpu() for k, v in model.state_dict().items() if "embedding_dict" not in k},
           "checksum": float(table.double().sum().item())}
    del ts, opt, model
    torch.cuda.empty_cache()
    return out


def _same(a, b, what):
    assert torch.equal(a["losses"], b["losses"]), f"{what}: losses differ"
    for k in ("rows", "m", "v"):
        assert torch.equal(a[k], b[k]), f"{what}: table {k} differs"
    for k in a["dense"]:
        assert torch.equal(a["dense"][k], b["dense"][k]), f"{what}: {k} differs"
    assert a["checksum"] == b["checksum"], f"{what}: whole-table checksum differs"


def test_full_size_step_is_deterministic_and_mode_independent(cuda):
    dense = _run(cuda, "dense")
    _same(dense, _run(cuda, "dense"), "dense, run twice")
    lazy = _run(cuda, "lazy")
    _same(dense, lazy, "lazy (exact replay) vs dense")
    _same(lazy, _run(cuda, "lazy", use_graph=True), "graph replay vs eager")
    # bf16 contractions + dropout (the bench configuration): still deterministic run to run, graph or not
    a = _run(cuda, "lazy", use_graph=True, fast_replay=True, precision="bf16", dropout=0.2)
    _same(a, _run(cuda, "lazy", use_graph=False, fast_replay=True, precision="bf16", dropout=0.2), "bench configuration, graph vs eager")
    # the row sort of the next batch one step ahead (what bench.py and data.train_epoch do): same bits
    _same(a, _run(cuda, "lazy", use_graph=True, fast_replay=True, precision="bf16", dropout=0.2, announce=True),
          "bench configuration, next batch announced vs every step sorting its own")
    # untouched rows moved ~lr per step, touched or not the table stayed finite
    assert np.isfinite(dense["checksum"])


def test_full_size_first_step_loss_and_gather_against_the_oracle(cuda):
    from cdcmdr_amd import _lib as L
    model = _model(cuda)
    X, y, g = _batches(1)
    Xd = torch.from_numpy(X[0]).to(cuda)
    model.train()
    pred = model(Xd)                                               # drop-in forward on the HIP plan
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = O.ple_forward(sd, X[0], [VOCAB] * FIELDS, 3, training=True)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-6)
    # gather: exact copy of the rows
    lib = L.load()
    table = model.embedding.embedding_dict.weight.detach()
    out = torch.empty((B, FIELDS * D), dtype=torch.float32, device=cuda)
    offs = model.embedding.offsets_device(cuda)
    L.check(lib.cdc_embed_gather_fwd(Xd.data_ptr(), offs.data_ptr(), table.data_ptr(), out.data_ptr(), None, None, B, FIELDS, D,
                                     table.shape[0], C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gather")
    rows = (Xd.long() + torch.arange(FIELDS, device=cuda) * VOCAB).reshape(-1)
    assert torch.equal(out.view(-1, D), table[rows])
