"""Evaluation path of the reference on the device: `Run.test` + `Run.evaluate_multi_domain`
(run.py:647-711).

The reference runs the model in eval mode batch by batch, moves every batch's predictions, labels and domain column to
the host (`.cpu().numpy()`, one synchronisation per batch) and calls sklearn's roc_auc_score / log_loss on the whole set
and per domain (pandas groupby).  Here the forward runs on the HIP plans (eval mode: BatchNorm on running statistics, no
dropout), predictions stay in HBM and ONE C-ABI call (`cdc_eval_metrics`) returns every figure; the only host
synchronisation is reading the result.
"""
import ctypes as C
import math

import torch

from . import _lib as L


def eval_metrics(pred, label, domain=None, n_domain=1):
    """pred f32 [n], label int16 [n] (0/1), domain int32 [n] or a strided column view (e.g. X[:, domain_idx]).
    Returns (auc, loss, rows, positives): device tensors with n_domain + 1 entries each — domains 0..n_domain-1, then ALL
    rows.  NaN where a segment is empty or holds a single class.  No host synchronisation."""
    lib = L.load()
    if not pred.is_cuda:
        raise L.HipExtensionError("eval_metrics needs device tensors; there is no CPU fallback")
    pred = pred.reshape(-1).to(torch.float32).contiguous()
    label = label.reshape(-1).to(torch.int16).contiguous()
    n = pred.numel()
    if label.numel() != n:
        raise ValueError(f"{n} predictions but {label.numel()} labels")
    ld = 0
    if domain is not None:
        if domain.dtype != torch.int32:
            domain = domain.to(torch.int32)
        domain = domain.reshape(-1) if domain.dim() > 1 and domain.is_contiguous() else domain
        if domain.dim() != 1 or domain.numel() != n:
            raise ValueError("domain must hold one entry per prediction")
        ld = domain.stride(0)
    elif n_domain != 1:
        raise ValueError("n_domain > 1 needs the domain column")
    dev = pred.device
    seg = n_domain + 1
    out = torch.empty(2 * seg, dtype=torch.float64, device=dev)
    counts = torch.empty(2 * seg, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = lib.cdc_eval_workspace_bytes(n, n_domain)
    if nbytes <= 0:
        raise RuntimeError("cdc_eval_workspace_bytes failed")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    L.launch("cdc_eval_metrics", lib.cdc_eval_metrics,
             (pred.data_ptr(), label.data_ptr(), None if domain is None else domain.data_ptr(), ld, n, n_domain, out.data_ptr(),
              counts.data_ptr(), err.data_ptr(), ws.data_ptr(), nbytes), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    eval_metrics.last_err = err
    return out[:seg], out[seg:], counts[:seg], counts[seg:]


class Evaluator:
    """Mirror of Run.test (run.py:647-690): the CDC, multi-tower and single-tower branches.

    mode "cdc": `data_loader` is ({domain: loader of (X, y)}, domain_batch_seq) as data.make_domain_loaders returns them for the
                validation / test split; for every d of the sequence the next batch of domain d is scored by the tower of d's
                group: pred = model(X, mode='split', domain_i=d)                       (run.py:653-661, get_domain_data 499-526)
    mode "multi": batches are (X, y, group) and pred = model(X).gather(1, group)   (run.py:668-673)
    mode "single": batches are (X, y) and pred = model(X)                           (run.py:674-676)
    domain_cnt_weight: {domain: weight} or a sequence, as Run.domain_cnt_weight (mean_auc / mean_loss, run.py:706-707)."""

    def __init__(self, model, mode="multi", domain_idx=None, n_domain=1, domain_cnt_weight=None, is_evaluate_multi_domain=True):
        self.model, self.mode = model, mode
        self.domain_idx, self.n_domain = domain_idx, int(n_domain)
        self.domain_cnt_weight = domain_cnt_weight
        self.is_evaluate_multi_domain = bool(is_evaluate_multi_domain) and domain_idx is not None

    def predict(self, data_loader):
        """-> (pred f32 [n], label int16 [n], domain int32 [n] or None), all on the device."""
        model = self.model
        was_training = model.training
        model.eval()
        preds, labels, domains = [], [], []
        try:
            with torch.no_grad():
                for batch in (self._domain_batches(*data_loader) if self.mode == "cdc" else data_loader):
                    if self.mode == "cdc":
                        d, X, y = batch
                        pred = model(X, mode='split', domain_i=d)
                    elif self.mode == "multi":
                        X, y, group = batch
                        pred = model(X).gather(1, group.reshape(-1, 1).to(torch.int64))
                    else:
                        X, y = batch
                        pred = model(X)
                    preds.append(pred.reshape(-1).to(torch.float32))
                    labels.append(y.reshape(-1).to(torch.int16))
                    if self.domain_idx is not None:
                        domains.append(X[:, self.domain_idx].to(torch.int32))
        finally:
            model.train(was_training)
        if not preds:
            raise ValueError("empty evaluation set")
        return torch.cat(preds), torch.cat(labels), (torch.cat(domains) if domains else None)

    @staticmethod
    def _domain_batches(loaders, domain_batch_seq):
        """Run.get_domain_data in 'valid' / 'test' mode (run.py:507-526): a generator per domain, restarted when exhausted; the
        sequence holds ceil(rows_d / bs) entries of d, so one pass visits every row of every domain once."""
        gens = {}
        for d in domain_batch_seq:
            d = int(d)
            if d not in gens:
                gens[d] = iter(loaders[d])
            try:
                X, y = next(gens[d])
            except StopIteration:
                gens[d] = iter(loaders[d])
                X, y = next(gens[d])
            yield d, X, y

    def test(self, data_loader):
        """The reference's result_dict: total_auc, total_loss (+ domain_auc, domain_loss, mean_auc, mean_loss)."""
        pred, label, domain = self.predict(data_loader)
        multi = self.is_evaluate_multi_domain
        auc, loss, rows, pos = eval_metrics(pred, label, domain if multi else None, self.n_domain if multi else 1)
        auc, loss, rows, pos = auc.cpu().tolist(), loss.cpu().tolist(), rows.cpu().tolist(), pos.cpu().tolist()   # the one sync
        bad = int(eval_metrics.last_err.item())
        if bad:
            raise ValueError(f"evaluation row {bad - 1}: NaN prediction, label outside {{0,1}} or domain outside [0, {self.n_domain})")
        if pos[-1] == 0 or pos[-1] == rows[-1]:
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")   # run.py:685
        result = {"total_auc": auc[-1], "total_loss": loss[-1]}
        if multi:
            domain_auc, domain_loss = {}, {}
            mean_auc, mean_loss = 0, 0
            for d in range(self.n_domain):
                if rows[d] == 0:
                    continue                                   # pandas groupby yields no group for an absent domain
                domain_auc[d], domain_loss[d] = auc[d], loss[d]
                w = self._weight(d)
                mean_auc += w * auc[d]
                mean_loss += w * loss[d]
            result.update({"domain_auc": domain_auc, "domain_loss": domain_loss, "mean_auc": mean_auc, "mean_loss": mean_loss})
        return result

    def _weight(self, d):
        w = self.domain_cnt_weight
        if w is None:
            return math.nan
        return float(w[d])
