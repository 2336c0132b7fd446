"""Data residency and batching of the reference: `Run.convert2data_loader` / `convert2domain_data_loader`
(run.py:208-293) and the epoch loop of `Run.train` (run.py:470-497).

The reference keeps the whole pre-processed dataset on the device (`int32[N,F]` ids, `int16[N,1]` labels, `int64[N,1]`
tower index) and iterates it with `DataLoader(TensorDataset(...), bs, shuffle=True)`: per batch, Python collates
`dataset[i]` for every index of the batch — about 4096 small device indexing ops per step, far more than the 0.9 ms the HIP
step itself takes.  `DeviceLoader` yields the SAME batches in the SAME order (it draws the epoch's permutation exactly like
torch's RandomSampler: a seed from the global generator, then `randperm` on a CPU generator) but forms each batch with one
device gather per tensor.  `rank`/`world` give the data-parallel variant: every rank walks the same permutation and takes
its contiguous share of each global batch of `world * batch_size` rows.

On-disk format: `{mode}_data_loader.pth` / `{mode}_label_loader.pth` as written by run.py:194-206 (plain `torch.save`d
tensors), read with `weights_only=True`.
"""
import os

import numpy as np
import torch


def load_split(folder, mode):
    """-> (X int32 [N,F], y int16 [N,1]) from the reference's pre-processed tensors (run.py:213-215)."""
    X = torch.load(os.path.join(folder, f"{mode}_data_loader.pth"), weights_only=True).to(torch.int32)
    y = torch.load(os.path.join(folder, f"{mode}_label_loader.pth"), weights_only=True).to(torch.int16)
    if X.dim() != 2 or y.shape[0] != X.shape[0]:
        raise ValueError(f"{mode}: ids {tuple(X.shape)} and labels {tuple(y.shape)} do not belong together")
    return X, y


def save_split(folder, mode, X, y):
    """the writer side of the same format (run.py:194-206)"""
    os.makedirs(folder, exist_ok=True)
    torch.save(X.to(torch.int32).cpu(), os.path.join(folder, f"{mode}_data_loader.pth"))
    torch.save(y.to(torch.int16).cpu().reshape(-1, 1), os.path.join(folder, f"{mode}_label_loader.pth"))


class DeviceLoader:
    """`DataLoader(TensorDataset(*tensors), batch_size, shuffle=shuffle)` over device-resident tensors, batch for batch."""

    def __init__(self, tensors, batch_size, shuffle=True, rank=0, world=1, generator=None):
        n = tensors[0].shape[0]
        if any(t.shape[0] != n for t in tensors):
            raise ValueError("Size mismatch between tensors")           # TensorDataset's own check
        if not (0 <= rank < world):
            raise ValueError("rank outside [0, world)")
        self.tensors, self.batch_size, self.shuffle = tuple(tensors), int(batch_size), bool(shuffle)
        self.rank, self.world, self.generator = int(rank), int(world), generator
        self.n = n
        self.dropped_rows = 0             # data parallel: rows of a last global batch with fewer rows than ranks (not trained on)
        self.last_global_rows = 0         # data parallel: true size of the ragged last global batch this epoch yielded shares of

    def __len__(self):
        g = self.batch_size * self.world
        if self.world > 1:                # the ragged last global batch is yielded when every rank gets at least one row of it
            return self.n // g + (1 if self.n % g >= self.world else 0)
        return (self.n + g - 1) // g

    def _order(self):
        if not self.shuffle:
            return None
        # What torch draws per epoch, in its order: the DataLoader iterator's base seed (unused by a worker-less loader, but
        # it advances the generator), then RandomSampler.__iter__: a seed from the global generator and randperm on a fresh
        # CPU generator — or randperm straight on the generator the loader was given.
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)
        if self.generator is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
        else:
            g = self.generator
        return torch.randperm(self.n, generator=g)

    def __iter__(self):
        perm = self._order()
        dev = self.tensors[0].device
        if perm is not None:
            perm = perm.to(dev)
        step = self.batch_size * self.world
        for lo in range(0, self.n, step):
            hi = min(lo + step, self.n)
            if self.world > 1:
                # the rank's contiguous share of the global batch.  The ragged last global batch (run.py:476 trains it like any other)
                # is split as evenly as its rows allow — rank r gets base + (r < rem) rows — and yielded with its true global size in
                # `last_global_rows`: train_epoch runs it through TrainStep.sibling(rows, global_rows=, cap_rows=), whose loss mean and
                # BatchNorm statistics are those of the whole ragged batch.  Fewer rows than ranks: not yielded (a rank without a row
                # cannot build a step), counted in dropped_rows
                if hi - lo < step:
                    n = hi - lo
                    if n < self.world:
                        self.dropped_rows = n
                        return
                    base, rem = divmod(n, self.world)
                    a = lo + self.rank * base + min(self.rank, rem)
                    b = a + base + (1 if self.rank < rem else 0)
                    self.last_global_rows = n
                else:
                    a, b = lo + self.rank * self.batch_size, lo + (self.rank + 1) * self.batch_size
            else:
                a, b = lo, hi
            if perm is None:
                yield tuple(t[a:b] for t in self.tensors)
            else:
                idx = perm[a:b]
                yield tuple(t.index_select(0, idx) for t in self.tensors)


def make_loader(X, y, batch_size, device, domain_idx=None, domain2group=None, domain_filter=None, shuffle=True, rank=0, world=1):
    """run.py:208-250 `convert2data_loader` after the tensors are loaded: optional domain filter, the tower index
    `group = domain2group[X[:, domain_idx]]` for multi-tower models, `domain_cnt_weight` (relative domain frequencies, used
    by evaluate_multi_domain), everything moved to `device`.  Returns (loader, domain_cnt_weight)."""
    X, y = X.to(torch.int32), y.to(torch.int16).reshape(-1, 1)
    if domain_filter is not None:
        mask = torch.isin(X[:, domain_idx], torch.as_tensor(list(domain_filter), dtype=X.dtype))
        X, y = X[mask], y[mask]
    weight = None
    if domain_idx is not None:
        cnt = X[:, domain_idx].to(torch.int64).bincount()
        weight = np.array([float(cnt[i]) / X.shape[0] for i in range(len(cnt))])
    tensors = [X.to(device), y.to(device)]
    if domain2group is not None:
        dom = X[:, domain_idx].to(torch.int64)
        if isinstance(domain2group, dict):
            table = torch.full((int(dom.max()) + 1,), -1, dtype=torch.int64)
            for d, g in domain2group.items():
                if 0 <= int(d) < table.numel():
                    table[int(d)] = int(g)
        else:
            table = torch.as_tensor(domain2group, dtype=torch.int64)
        group = table[dom]
        if int(group.min()) < 0:
            raise ValueError("a domain of the data has no tower in domain2group")   # pandas .map would give NaN -> cast error
        tensors.append(group.view(-1, 1).to(device))
    return DeviceLoader(tensors, batch_size, shuffle=shuffle, rank=rank, world=world), weight


def make_domain_loaders(X, y, batch_size, device, domain_idx, n_domain, domain_filter=None, shuffle=True):
    """run.py:252-293 `convert2domain_data_loader`: one loader per domain plus the shuffled sequence saying which domain
    each batch of an epoch comes from (`np.random.shuffle`, the reference's generator).
    Returns (loaders {domain: DeviceLoader}, batch_seq list, domain_cnt_weight)."""
    X, y = X.to(torch.int32), y.to(torch.int16).reshape(-1, 1)
    if domain_filter is not None:
        mask = torch.isin(X[:, domain_idx], torch.as_tensor(list(domain_filter), dtype=X.dtype))
        X, y = X[mask], y[mask]
    loaders, seq = {}, []
    for d in (domain_filter if domain_filter is not None else range(n_domain)):
        mask = X[:, domain_idx] == d
        dX, dy = X[mask].to(device), y[mask].to(device)
        loaders[d] = DeviceLoader((dX, dy), batch_size, shuffle=shuffle)
        seq.extend([d] * int(np.ceil(dX.shape[0] * 1.0 / batch_size)))
    cnt = X[:, domain_idx].to(torch.int64).bincount()
    weight = np.array([float(cnt[i]) / X.shape[0] for i in range(len(cnt))])
    np.random.shuffle(seq)
    return loaders, seq, weight


def train_epoch(step, loader, log_interval=None, log=None):
    """`Run.train` (run.py:470-497) on a TrainStep: one pass over the loader; every `log_interval` batches (reference:
    204800 // bs) the mean of loss + regularisation term is reported through `log(mean)` — the only host synchronisation.
    The ragged last batch of an epoch is trained on like any other (a sibling step of that size on the same model and
    optimiser state); under data parallelism every rank gets its share of it (DeviceLoader) and the sibling step is told the
    batch's global size: loss mean, BatchNorm statistics and row-list capacities are those of the whole ragged batch.
    LOADER CONTRACT: the loop looks one batch ahead (the next batch's row sort runs beside this batch's step), so a yielded batch
    must stay valid and unchanged while the NEXT one is produced and until its own step has been issued — DeviceLoader yields
    fresh tensors or views of the resident data; a loader that refills ONE staging buffer per batch must double-buffer.  The
    look-ahead recognises "the batch I was told about" by (address, tensor version): contents changed behind torch's back
    (`.data.copy_`, a custom kernel, DLPack) are not seen.
    Out-of-range ids and row-list overflows are surfaced at the logging synchronisations and at the end of the epoch
    (`step.check_ids()`: IndexError like nn.Embedding's).  Returns (batches run, batches skipped)."""
    if log_interval is None:
        log_interval = max(1, 204800 // step.B)
    acc = torch.zeros((), dtype=torch.float64, device=step.device)
    done = skipped = 0
    step.refresh_table_reg()                     # lazy table: the first logging window reports the table's L2 term too
    it = iter(loader)
    batch = next(it, None)
    while batch is not None:
        nxt = next(it, None)                     # one batch of look-ahead: its rows are sorted beside this batch's step
        X = batch[0]
        ts = step
        if X.shape[0] != step.B:
            if X.shape[0] == 0:
                skipped += 1
                batch = nxt
                continue
            if step.world > 1:
                n = int(getattr(loader, "last_global_rows", 0))
                assert n > 0, "a ragged local batch under data parallelism needs the loader's last_global_rows"
                ts = step.sibling(X.shape[0], global_rows=n, cap_rows=-(-n // step.world))
            else:
                ts = step.sibling(X.shape[0])
        ahead = nxt[0] if (nxt is not None and ts is step and nxt[0].shape[0] == step.B) else None
        bce, reg = ts.step(*batch, next_X=ahead) if ahead is not None else ts.step(*batch)
        batch = nxt
        acc += bce.double().sum() + reg
        done += 1
        if done % log_interval == 0:
            if log is not None:
                log(float(acc.item()) / log_interval)
            acc.zero_()
            ts.check_ids()                       # the host is synchronised here anyway
            step.refresh_table_reg()             # lazy table: the reported loss's table term, exact again from here
    step.check_ids()
    for sib in step.__dict__.get("_siblings", {}).values():
        sib.check_ids()
    if getattr(loader, "dropped_rows", 0):
        skipped += 1
    return done, skipped
