"""The CDC training loop on the HIP path: `Run.train_cdc` (run.py:596-645), `Run.update_matrix_cdc`
(run.py:528-594) and `Run.get_domain_data` (run.py:499-526).

What the reference does per epoch: (first epoch) `warmup_step` steps on single-domain batches with the tower MEAN as the
prediction; then one step per entry of the shuffled domain batch sequence on that domain's tower, and — before the very
first of those and then every `update_interval` steps — a re-estimation of the affinity matrices: snapshot the base model,
and for each of n_causal_mask random domain subsets / each domain / each (domain's source set | cluster) train
`update_matrix_step` steps, evaluate every domain's metric on one of its training batches, restore the snapshot; finally
`CDC.update_group()` regroups the domains (clustering.py).

Mirrored quirk: the evaluation pass puts the model in eval mode (run.py:550) and only the matrix-A loop switches back to
train mode (run.py:571); every other training step after the first evaluation — the rest of the matrix update AND the
remainder of the epoch — runs with BatchNorm on its running statistics and without dropout.  `self.training` tracks the
module flag exactly as the reference's calls set it, and picks the matching TrainStep.

The reference's driver class cannot be imported here (it needs the absent `dataset` package and wandb), so this loop is
"parity unpinned" as a whole; its parts are pinned separately: the step (tests/test_gpu_train.py), the eval forward and
metrics (tests/test_gpu_eval.py), the regrouping arithmetic (tests/test_cdc_group.py, against the reference's outputs).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .evaluate import eval_metrics
from .trainer import TrainStep


class CDCTrainer:
    MAX_STEPS = 16      # resident TrainSteps (one per batch size x flavour); least recently used ones are rebuilt on demand

    def __init__(self, model, optimizer, batch_size, train_loaders, n_domain, domain_cnt_weight, train_domain_batch_seq,
                 warmup_step=200, update_matrix_step=2, update_interval=1000, n_causal_mask=None, log=None):
        """model: cdcmdr_amd.model.cdc.CDC on the GPU; optimizer: FusedAdam(model.base_model_instance);
        train_loaders {domain: DeviceLoader of (X, y)} and train_domain_batch_seq from data.make_domain_loaders;
        the three step counts are the reference's config values (config.py:57-59), rescaled like run.py:601-604."""
        self.model, self.opt, self.bs = model, optimizer, int(batch_size)
        self.base = model.base_model_instance
        self.loaders = train_loaders
        self.n_domain, self.n_cluster = int(n_domain), model.n_cluster
        self.domain_cnt_weight = np.asarray(domain_cnt_weight, dtype=np.float64)
        self.seq = list(train_domain_batch_seq)
        self.warmup_step = max(5, (warmup_step * 1024) // self.bs)
        self.update_matrix_step = max(1, (update_matrix_step * 1024) // self.bs) if update_matrix_step != 0 else 0
        self.update_interval = (update_interval * 1024) // self.bs
        self.n_causal_mask = model.n_causal_mask if n_causal_mask is None else int(n_causal_mask)
        self.log = log
        self.device = optimizer.device
        self.training = True                              # the module flag as the reference's calls leave it
        self._gen = {}
        self._steps = {}
        self.domain2group_list = list(model.domain2group_list)

    # ---- run.py:499-526 ------------------------------------------------------------------------------------
    def get_domain_data(self, d):
        """one training batch of domain d (its loader restarts when exhausted), or the concatenation of one batch of every
        domain of a list (which is shuffled in place, like the reference does)"""
        if isinstance(d, (int, np.integer)):
            d = int(d)
            it = self._gen.get(d)
            if it is None:
                it = self._gen[d] = iter(self.loaders[d])
            try:
                return next(it)
            except StopIteration:
                it = self._gen[d] = iter(self.loaders[d])
                return next(it)
        np.random.shuffle(d)
        parts = [self.get_domain_data(di) for di in d]
        return torch.cat([p[0] for p in parts], dim=0), torch.cat([p[1] for p in parts], dim=0)

    # ---- one optimisation step in the reference's three forward modes ----------------------------------------
    def _step(self, X, y, mode, domain_i=None):
        B = X.shape[0]
        kind = "mean" if mode == "warmup" else "multi"
        key = (kind, B, self.training)
        ts = self._steps.pop(key, None)
        if ts is None:
            ts = TrainStep(self.base, self.opt, B, mode=kind, use_graph=False, train_mode=self.training)
        self._steps[key] = ts                                         # most recently used last
        while len(self._steps) > self.MAX_STEPS:                      # every step owns its batch size's buffers
            self._steps.pop(next(iter(self._steps)))
        if kind == "mean":
            group = None
        elif domain_i is not None:                                    # cdc.py:108-111: one tower for the whole batch
            group = torch.full((B,), int(self.model.domain2group_list[domain_i]), dtype=torch.int64, device=X.device)
        else:                                                         # cdc.py:104-107: every row's own domain's tower
            group = self.model.groups_of(X)
        return ts.step(X, y, group)

    # ---- run.py:528-594 ------------------------------------------------------------------------------------
    def _train_with(self, domains, n_interval):
        if isinstance(domains, (int, np.integer)):
            todo = [int(domains)] * n_interval
        else:
            flat = list(domains) * n_interval
            todo = [flat[i:i + 7] for i in range(0, len(flat), 7)]
        for item in todo:
            X, y = self.get_domain_data(item)
            if isinstance(item, int):
                self._step(X, y, "split", domain_i=item)
            else:
                self._step(X, y, "split")

    def _test_all_domains(self):
        """every domain's metric on one of its training batches, eval-mode forward (run.py:549-558)"""
        self.training = False
        self.model.eval()
        self.opt.flush_table()                                        # the eval forward reads the table directly
        row = torch.zeros(self.n_domain, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for d in range(self.n_domain):
                X, y = self.get_domain_data(d)
                pred = self.base(X)[:, int(self.model.domain2group_list[d])]
                if self.model.use_metric == 'loss':
                    row[d] = F.binary_cross_entropy(pred, y.reshape(-1).float())
                else:
                    auc, _, _, _ = eval_metrics(pred, y.reshape(-1))
                    row[d] = auc[-1].float()
        return row

    def _snapshot(self):
        self.opt.flush_table()
        self.model.save_model_state()

    def _restore(self):
        # the reference restores the WEIGHTS only (cdc.py:351-354): the Adam moments of every row keep the k probe steps they
        # have lived through under its dense optimiser.  Lazy table: replay those steps for the rows that were not looked up
        # (flush) BEFORE the weights are overwritten, then mark every row current.
        self.opt.flush_table()
        self.model.load_model_state()
        if self.opt.table_mode == "lazy":                             # the restored rows ARE the current values
            self.opt.table_last.fill_(int(self.opt.step_dev.item()))

    def update_matrix(self):
        m, n, k = self.model, self.n_domain, self.update_matrix_step
        self._snapshot()
        for line in range(self.n_causal_mask):                        # treatment matrix: random domain subsets
            subset = np.random.choice(range(n), p=self.domain_cnt_weight, size=np.random.randint(5, n))
            self._train_with(subset, k)
            m.matrix_mask[line] = self._test_all_domains()
            self._restore()
        m.matrix_A[n] = self._test_all_domains()                      # the warmed-up model alone
        for d in range(n):                                            # matrix A: train on one domain
            self.training = True
            self.model.train()
            self._train_with(d, k)
            m.matrix_A[d] = self._test_all_domains()
            self._restore()
        rows = n + self.n_cluster if max(m.domain2group_list) > 0 else n + 1
        for r in range(rows):                                         # matrix B: train on the source set without d / the cluster
            if r >= n:
                # run.py:584 takes domain2group_list[r - n] — the CLUSTER INDEX of domain (r - n), a scalar — as the
                # training "domain"; mirrored as written
                domains = int(m.domain2group_list[r - n])
            else:
                domains = [d for d in m.s_group2domain_list[m.domain2group_list[r]] if d != r]
            self._train_with(domains, k)
            m.matrix_B[r] = self._test_all_domains()
            self._restore()
        self.domain2group_list = m.update_group()
        return self.domain2group_list

    # ---- run.py:596-645 ------------------------------------------------------------------------------------
    def train_epoch(self, epoch_i):
        self.training = True
        self.model.train()
        log_interval = max(1, 204800 // self.bs)
        acc, seen = 0.0, 0

        def account(bce, reg):
            nonlocal acc, seen
            acc = acc + bce.double().sum() + reg
            seen += 1
            if seen % log_interval == 0:
                if self.log is not None:
                    self.log(float(acc) / log_interval)
                acc = 0.0
                self.check_ids()
                self.opt.refresh_table_reg()                          # lazy table: the reported loss's table term (run.py:637), exact again

        self.opt.refresh_table_reg()

        if epoch_i == 0:                                              # warm-up on the tower mean
            for _ in range(self.warmup_step):
                d = np.random.choice(range(self.n_domain), p=self.domain_cnt_weight)
                X, y = self.get_domain_data(d)
                account(*self._step(X, y, "warmup"))
            acc, seen = 0.0, 0
        for i, d in enumerate(self.seq):
            X, y = self.get_domain_data(d)
            if (epoch_i == 0 and i == 0) or (self.update_interval > 0 and (i + 1) % self.update_interval == 0):
                self.update_matrix()
            account(*self._step(X, y, "split", domain_i=int(d)))
        self.check_ids()
        return len(self.seq)

    def check_ids(self):
        """host-synchronising: IndexError if any resident step has seen an out-of-range id since the last check"""
        for ts in list(self._steps.values()):
            ts.check_ids()
