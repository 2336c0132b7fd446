"""The epoch loop of the reference's driver on the HIP path: `Run.main` (run.py:713-770) with `Run.is_continuable`
(run.py:440-468) — train an epoch, evaluate on the validation set, keep the best checkpoint by `mean_auc` (falling back to
`total_auc` when no per-domain evaluation is configured), stop after `num_trials` epochs without improvement, reload the best
checkpoint and evaluate on the test set.  wandb logging, dataset preprocessing and model construction from `config` stay
with the caller (they are the reference's control plane: SURVEY §2)."""
import torch

from .data import train_epoch
from .evaluate import Evaluator


class Runner:
    def __init__(self, model, step, evaluator: Evaluator, save_model_path, num_trials=3, cdc_trainer=None, log=print):
        """step: the TrainStep (its optimiser is checkpointed); cdc_trainer: a CDCTrainer whose train_epoch replaces the
        plain epoch for CDC models (run.py:738-750)."""
        self.model, self.step, self.evaluator = model, step, evaluator
        self.save_model_path, self.num_trials = save_model_path, int(num_trials)
        self.cdc_trainer, self.log = cdc_trainer, log
        self.best_auc = self.best_loss = self.best_mean_auc = self.best_mean_loss = 0.0
        self.trial_counter = 0

    def _sync_table(self):
        """every table row at the current step (and on every rank) before the parameters are read"""
        if getattr(self.step, "world", 1) > 1:
            self.step.gather_table()
        else:
            self.step.opt.flush_table()

    def _rank(self):
        d = getattr(self.step, "dist", None)
        return 0 if d is None else int(getattr(d, "rank", 0))

    def _barrier(self):
        d = getattr(self.step, "dist", None)
        if d is not None and getattr(self.step, "world", 1) > 1:
            d.barrier()

    def is_continuable(self, result_dict, epoch_i):
        score = result_dict.get("mean_auc")
        best = self.best_mean_auc
        if score is None:                       # is_evaluate_multi_domain off: the reference's commented-out total_auc criterion
            score, best = result_dict["total_auc"], self.best_auc
        if score > best:
            self.trial_counter = 0
            self.best_auc, self.best_loss = result_dict["total_auc"], result_dict["total_loss"]
            save = {"epoch": epoch_i + 1, "state_dict": self.model.state_dict(), "best_auc": self.best_auc,
                    "best_result": result_dict, "optimizer": self.step.opt.state_dict()}
            if result_dict.get("mean_auc") is not None:
                self.best_mean_auc, self.best_mean_loss = result_dict["mean_auc"], result_dict["mean_loss"]
                save["best_mean_auc"], save["best_mean_loss"] = self.best_mean_auc, self.best_mean_loss
            if hasattr(self.model, "domain2group_list"):
                save["domain2group_list"] = [int(v) for v in self.model.domain2group_list]
                save["s_group2domain_list"] = [[int(v) for v in g] for g in self.model.s_group2domain_list]
            if self._rank() == 0:                # one writer; the other ranks hold the same state after _sync_table()
                torch.save(save, self.save_model_path)
            self.log(f"current best epoch: {epoch_i + 1}, auc: {self.best_auc:.4f}, loss: {self.best_loss:.4f}")
            return True
        if self.trial_counter + 1 < self.num_trials:
            self.trial_counter += 1
            return True
        return False

    def fit(self, train_loader, valid_loader, test_loader=None, epochs=1):
        epoch_i = -1
        for epoch_i in range(epochs):
            if self.cdc_trainer is not None:
                self.cdc_trainer.train_epoch(epoch_i)
            else:
                self.model.train()
                train_epoch(self.step, train_loader)
            self._sync_table()
            result = self.evaluator.test(valid_loader)
            self.log(f"validation: auc: {result['total_auc']:.4f}, loss: {result['total_loss']:.4f}")
            if not self.is_continuable(result, epoch_i):
                break
        self._barrier()                          # rank 0's last save is complete before anyone reads it
        ck = torch.load(self.save_model_path, map_location=self.step.device, weights_only=False)     # our own file (result dicts inside)
        self.model.load_state_dict(ck["state_dict"])
        self.step.opt.load_state_dict(ck["optimizer"])
        out = {"epochs_run": epoch_i + 1, "best": ck["best_result"]}
        if test_loader is not None:
            out["test"] = self.evaluator.test(test_loader)
        return out
