"""SURVEY §8f N3 on CPU: DeviceLoader yields exactly the batches torch's own
DataLoader(TensorDataset(...), bs, shuffle=True) yields for the same RNG state (what run.py:244 builds), the reference's
on-disk tensor format round-trips, tower indices / domain weights follow run.py:228-237, and the rank-sharded variant
partitions every global batch."""
import os
import sys

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _data(n=1000, F=5, seed=0):
    rng = np.random.default_rng(seed)
    X = torch.from_numpy(rng.integers(0, 50, size=(n, F)).astype(np.int32))
    X[:, 2] = torch.from_numpy(rng.integers(0, 4, size=n).astype(np.int32))
    y = torch.from_numpy(rng.integers(0, 2, size=(n, 1)).astype(np.int16))
    return X, y


@pytest.mark.parametrize("bs", [64, 1000, 333])
def test_device_loader_equals_torch_dataloader_batch_for_batch(bs):
    from cdcmdr_amd.data import DeviceLoader
    X, y = _data()
    g = X[:, 2:3].to(torch.int64)
    torch.manual_seed(123)
    ref = DataLoader(TensorDataset(X, y, g), bs, shuffle=True)
    want = [tuple(t.clone() for t in b) for _ in range(2) for b in ref]            # two epochs: RNG advances like the sampler's
    torch.manual_seed(123)
    mine = DeviceLoader((X, y, g), bs, shuffle=True)
    got = [b for _ in range(2) for b in mine]
    assert len(mine) == len(ref) and len(got) == len(want)
    for a, b in zip(got, want):
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    # shuffle=False: plain slices
    for a, b in zip(DeviceLoader((X, y), bs, shuffle=False), DataLoader(TensorDataset(X, y), bs, shuffle=False)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_rank_shards_partition_every_global_batch():
    from cdcmdr_amd.data import DeviceLoader
    X, y = _data(n=1003)
    world, bs = 4, 50
    torch.manual_seed(7)
    whole = [b for b in DeviceLoader((X, y), bs * world, shuffle=True)]
    parts = []
    for r in range(world):
        torch.manual_seed(7)
        parts.append([b for b in DeviceLoader((X, y), bs, shuffle=True, rank=r, world=world)])
    # 1003 rows = 5 whole global batches of 200 + 3 ragged rows, which NO rank yields
    assert all(len(p) == len(whole) - 1 == 5 for p in parts)
    for i, b in enumerate(whole[:-1]):
        cat = torch.cat([parts[r][i][0] for r in range(world)])
        assert torch.equal(cat, b[0])                            # rank order == batch order (the DP parity definition)
    assert all(b[0].shape[0] == bs for p in parts for b in p)


@pytest.mark.parametrize("n", [2 * 3 * 64 - 1, 2 * 3 * 64 + 3 * 64 - 1, 7])
def test_every_rank_runs_the_same_number_of_batches(n):
    """Every rank must run the same number of steps (a rank that runs a step issues collectives the others have to join): the
    ragged last GLOBAL batch — here one row short of world*B, the case where ceil-splitting gave every rank but the last a full
    batch — is shared out over ALL ranks (round 4; round 3 skipped it on all ranks), unless it has fewer rows than there are ranks."""
    from cdcmdr_amd.data import DeviceLoader
    X, y = _data(n=n)
    world, bs = 3, 64
    counts = []
    for r in range(world):
        torch.manual_seed(3)
        dl = DeviceLoader((X, y), bs, shuffle=True, rank=r, world=world)
        batches = list(dl)
        tail = n % (world * bs)
        full = n // (world * bs)
        assert all(b[0].shape[0] == bs for b in batches[:full])
        assert len(batches) == len(dl) == full + (1 if tail >= world else 0)
        if tail >= world:
            assert dl.last_global_rows == tail and batches[-1][0].shape[0] in (tail // world, tail // world + 1)
        else:
            assert dl.dropped_rows == tail
        counts.append((len(batches), sum(b[0].shape[0] for b in batches)))
    assert len(set(c for c, _ in counts)) == 1
    assert sum(r for _, r in counts) == n - (n % (world * bs) if n % (world * bs) < world else 0)


def test_split_files_round_trip_and_make_loader(tmp_path):
    from cdcmdr_amd.data import load_split, make_domain_loaders, make_loader, save_split
    X, y = _data()
    save_split(str(tmp_path), "train", X, y)
    X2, y2 = load_split(str(tmp_path), "train")
    assert X2.dtype == torch.int32 and y2.dtype == torch.int16 and torch.equal(X2, X) and torch.equal(y2, y)
    d2g = {0: 0, 1: 1, 2: 1, 3: 2}
    loader, w = make_loader(X, y, 128, "cpu", domain_idx=2, domain2group=d2g, shuffle=False)
    cnt = np.bincount(X[:, 2].numpy())
    np.testing.assert_allclose(w, cnt / len(X))                  # run.py:233-237
    for xb, yb, gb in loader:
        assert gb.dtype == torch.int64 and gb.shape == (xb.shape[0], 1)
        assert gb.view(-1).tolist() == [d2g[int(d)] for d in xb[:, 2]]
    # domain filter (run.py:223-226) + per-domain loaders with the batch sequence (run.py:252-293)
    loader, _ = make_loader(X, y, 128, "cpu", domain_idx=2, domain_filter=[1, 3], shuffle=False)
    assert all(set(xb[:, 2].tolist()) <= {1, 3} for xb, _ in loader)
    np.random.seed(0)
    loaders, seq, w2 = make_domain_loaders(X, y, 100, "cpu", 2, 4)
    assert sorted(loaders) == [0, 1, 2, 3]
    for d, ld in loaders.items():
        assert seq.count(d) == len(ld) == int(np.ceil(cnt[d] / 100))
        assert all(bool((xb[:, 2] == d).all()) for xb, _ in ld)
    np.testing.assert_allclose(w2, cnt / len(X))
    with pytest.raises(ValueError):
        make_loader(X, y, 128, "cpu", domain_idx=2, domain2group={0: 0, 1: 1})     # domains 2, 3 have no tower


def test_ragged_last_global_batch_is_split_over_the_ranks():
    """Data parallel: the ragged last global batch (run.py:476 trains it) reaches every rank — shares differ by at most one row, cover
    the batch exactly once in order, and the loader reports its true global size; with fewer rows than ranks it is dropped."""
    from cdcmdr_amd.data import DeviceLoader
    n, bs = 75, 16
    X = torch.arange(n, dtype=torch.int32).reshape(-1, 1)
    y = torch.zeros(n, 1, dtype=torch.int16)
    for world in (2, 3, 4):
        got, last = [], None
        per_rank = []
        for rank in range(world):
            ld = DeviceLoader((X, y), bs, shuffle=False, rank=rank, world=world)
            batches = [b[0].reshape(-1).tolist() for b in ld]
            per_rank.append(batches)
            last = getattr(ld, "last_global_rows", None)
        full = n // (bs * world)
        tail = n - full * bs * world
        assert all(len(b) == full + (1 if tail >= world else 0) for b in per_rank)
        if tail >= world:
            assert last == tail
            shares = [b[-1] for b in per_rank]
            assert max(map(len, shares)) - min(map(len, shares)) <= 1
            assert sum(shares, []) == list(range(full * bs * world, n))
        for step in range(full):
            rows = sum((per_rank[r][step] for r in range(world)), [])
            assert rows == list(range(step * bs * world, (step + 1) * bs * world))
    ld = DeviceLoader((X[:66], y[:66]), 16, shuffle=False, rank=0, world=4)     # 66 = 64 + 2 rows: fewer rows than ranks in the tail
    assert len(list(ld)) == 1 and ld.dropped_rows == 2
