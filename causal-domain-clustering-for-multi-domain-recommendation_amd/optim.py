"""Adam with the reference's exact semantics, fused for the HIP path.

The reference trains with torch.optim.Adam(lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd) over ALL parameters
(run.py:720-721) and adds sum(l2 * w^2) over the registered tensors — the whole embedding table included — to the
loss every step (run.py:489, model/layer.py:31,96-112).  So every table row moves every step, looked up or not
(SURVEY.md F3).  This optimiser reproduces that, in two interchangeable forms for the table:

  table_mode="dense"  touched rows (batch gradient + L2) -> side buffer, one streaming pass over the whole
                      table for the L2-only rows, then the side buffer is patched in.  12 B read + 12 B
                      written per table element per step: the HBM-roofline kernel of the step.
  table_mode="lazy"   rows carry the step they are valid for; the L2-only recurrence is replayed only when a
                      row is next looked up (before the gather) or when flush() is called (state_dict, eval,
                      regularisation-loss report).  Bit-identical to "dense" (same per-element routine).

Dense parameters go through one multi-tensor kernel with the L2 gradient 2*l2*w folded in.
"""
import ctypes as C
import math

import os

import torch

from . import _lib as L


def step_scalar_table(lr, beta1, beta2, n=4096, truncate=False):
    """[n,2] fp32: step_size_t = lr / (1 - beta1^t) and sqrt(1 - beta2^t), formed in double like torch/optim/adam.py.
    truncate: cut the table right after both columns have reached their fp32 limits (kernels clamp the index)."""
    rows = [(0.0, 1.0)]
    for t in range(1, n):
        rows.append((lr / (1.0 - beta1 ** t), math.sqrt(1.0 - beta2 ** t)))
    tab = torch.tensor(rows, dtype=torch.float64).to(torch.float32)
    if truncate:
        same = (tab == tab[-1]).all(dim=1)
        first = int(torch.nonzero(~same).max().item()) + 1 if bool((~same).any()) else 0
        tab = tab[:first + 1].contiguous()
    return tab


class FusedAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, table_mode="dense", frozen=(),
                 fast_replay=True, flush_every=64):
        """frozen: parameters left untouched by the optimiser (like leaving them out of torch.optim.Adam's list).
        lazy mode: fast_replay = hardware rcp/sqrt in the replay of L2-only steps (False: the exact routine, bit-identical
        to the dense mode); flush_every = P: every step replays one of P slices of the table, so that every row is brought
        up to date once per P steps (bounds the replay gaps; 0 = only on demand)."""
        assert table_mode in ("dense", "lazy")
        self.model = model
        self.frozen = {id(p) for p in frozen}
        self.lib = L.load()
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.table_mode = table_mode
        self.table = model.embedding.embedding_dict.weight
        dev = self.table.device
        if dev.type != "cuda":
            raise L.HipExtensionError("FusedAdam needs the model on a GPU; there is no CPU fallback")
        self.device = dev
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)       # 1-based step of the current update
        tab = step_scalar_table(lr, betas[0], betas[1], n=65536, truncate=True)
        self.scalars = tab.to(dev).contiguous()
        self.inv_bc2 = (1.0 / tab[:, 1].double()).to(torch.float32).to(dev).contiguous()
        self.fast_replay = bool(fast_replay) and table_mode == "lazy"
        self.flush_every = int(flush_every) if table_mode == "lazy" else 0
        self.reg_sum = torch.zeros(2, dtype=torch.float64, device=dev)       # [0] dense params (l2 applied), [1] table sum(w^2)
        self.table_reg = torch.zeros((), dtype=torch.float64, device=dev)    # lazy table: l2 * sum(w^2) at the last refresh
        self.table_reg_ready = False          # False until refresh_table_reg() has run for the current weights (build / load)
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float64).to(torch.float32))  # noqa: E731
        self._lerp_w = f32(1.0 - betas[0])
        self._beta2 = f32(betas[1])
        self._omb2 = f32(1.0 - betas[1])
        self._eps = f32(eps)
        self._wd = f32(weight_decay)
        # per-parameter L2 coefficient from the model's regularisation registry
        l2_of = {}
        for p, l1, l2 in model.regularized_parameters():
            if l1:
                raise NotImplementedError("l1 regularisation is never enabled by the reference's models")
            l2_of[id(p)] = l2_of.get(id(p), 0.0) + l2
        self._l2_of = l2_of
        self.l2_table = l2_of.get(id(self.table), 0.0)
        # fast replay in scaled state (csrc/common.h adam_replay_wave_scaled): constants and the per-step table, in double
        self._k1 = self._k2 = 0.0
        self.replay_tab = None
        c = float(torch.tensor(2.0 * f32(self.l2_table) + self._wd, dtype=torch.float64).to(torch.float32))
        self._ik1 = self._ik2 = self._k1_lo = self._k2_lo = 0.0
        if self.fast_replay and c > 0.0:
            k1, k2 = f32(self._lerp_w * c), f32(self._omb2 * c * c)
            if k1 > 0.0 and k2 > 0.0:
                # the scales as the kernel applies them: in by the fp32 reciprocals ik, out by K = 1 / ik (as fp32 pair hi + lo), the
                # per-step constants formed with the same K — in and out are inverses to 2^-48 (cdc_adam_hp)
                self._ik1, self._ik2 = f32(1.0 / k1), f32(1.0 / k2)
                K1, K2 = 1.0 / self._ik1, 1.0 / self._ik2
                self._k1, self._k2 = f32(K1), f32(K2)
                self._k1_lo, self._k2_lo = f32(K1 - self._k1), f32(K2 - self._k2)
                t64 = tab.double()
                A = t64[:, 0] * K1 * t64[:, 1] / math.sqrt(K2)
                E = self._eps * t64[:, 1] / math.sqrt(K2)
                A[0] = A[1]                                           # (row 0 = "step 0" is never replayed; keep it finite)
                rt = torch.stack([-1.0 / A, -E / A], dim=1)           # C1 = -1/A_t, C2 = -E_t/A_t (csrc/common.h adam_scaled_step_pk)
                self.replay_tab = rt.to(torch.float32).to(dev).contiguous()
        self.state = {}                                                       # id(param) -> (m, v)
        self.table_m = torch.zeros_like(self.table.data)
        self.table_v = torch.zeros_like(self.table.data)
        self.table_last = torch.zeros(self.table.shape[0], dtype=torch.int32, device=dev) if table_mode == "lazy" else None
        self.own_mod, self.own_rem = 0, 0     # row-sharded table (trainer sets them): this rank maintains rows r % own_mod == own_rem
        self._dense_args = None
        self._dense_sig = None
        self._ws = {}
        total_dense = sum(p.numel() + 4 for p in model.parameters() if p is not self.table)
        self.grad_arena = torch.zeros(total_dense, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0

    # ------------------------------------------------------------------------------------------
    def _hp(self):
        hp = L.AdamHP()
        hp.lerp_w, hp.beta2, hp.one_minus_beta2, hp.eps = self._lerp_w, self._beta2, self._omb2, self._eps
        hp.weight_decay = self._wd
        hp.l2_twice = 2.0 * float(torch.tensor(self.l2_table, dtype=torch.float64).to(torch.float32))
        hp.step_scalars = self.scalars.data_ptr()
        hp.n_scalars = self.scalars.shape[0]
        hp.fast_replay = 1 if self.fast_replay else 0
        hp.inv_bc2 = self.inv_bc2.data_ptr()
        hp.replay_tab = None if self.replay_tab is None else self.replay_tab.data_ptr()
        hp.k1, hp.k2 = self._k1, self._k2
        hp.ik1, hp.ik2, hp.k1_lo, hp.k2_lo = self._ik1, self._ik2, self._k1_lo, self._k2_lo
        return hp

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _workspace(self, B, F, D, tag=""):
        key = (B, F, D, tag)
        ws = self._ws.get(key)
        if ws is None:
            dev = self.device
            ws = {"uniq": torch.empty((F, B), dtype=torch.int32, device=dev),
                  "seg": torch.empty((F, B + 1), dtype=torch.int32, device=dev),
                  "perm": torch.empty((F, B), dtype=torch.int32, device=dev),
                  "cnt": torch.zeros((F,), dtype=torch.int32, device=dev),
                  "scratch": torch.empty((2 * F * B,), dtype=torch.int64, device=dev) if B > 1024 else None,
                  "rowgrad": torch.empty((F * B * D,), dtype=torch.float32, device=dev),
                  "side": torch.empty((F * B * 3 * D,), dtype=torch.float32, device=dev) if self.table_mode == "dense" else None}
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------------------------------
    def begin_step(self):
        """++step and clear the regularisation accumulators (call before the forward of the step)."""
        s = self._stream()
        L.launch("cdc_begin_step", self.lib.cdc_begin_step, (self.step_dev.data_ptr(), self.reg_sum.data_ptr(), 2), s)

    def sort_rows(self, idx, B, F, D, tag="", runs=0):
        """runs > 0: idx consists of `runs` runs that are already ascending per field (an owner's received row lists)."""
        ws = self._workspace(B, F, D, tag)
        if runs > 0:
            if ws.get("merge") is None:
                ws["merge"] = torch.empty((F * B,), dtype=torch.int64, device=self.device)
            L.launch("cdc_embed_merge_dedupe", self.lib.cdc_embed_merge_dedupe,
                     (idx.data_ptr(), ws["uniq"].data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), ws["cnt"].data_ptr(),
                      ws["merge"].data_ptr(), B, F, runs), self._stream())
            return ws
        L.launch("cdc_embed_sort_dedupe", self.lib.cdc_embed_sort_dedupe,
                 (idx.data_ptr(), ws["uniq"].data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), ws["cnt"].data_ptr(),
                  None if ws["scratch"] is None else ws["scratch"].data_ptr(), B, F), self._stream())
        return ws

    def table_index(self, ids, offsets, idx, B, F, err=None):
        """row indices of the local batch (needed before the gather in lazy mode, and for the exchanges under DP)."""
        L.launch("cdc_embed_index", self.lib.cdc_embed_index,
                 (ids.data_ptr(), offsets.data_ptr(), idx.data_ptr(), None if err is None else err.data_ptr(), B, F,
                  self.table.shape[0]), self._stream())

    def flush_slice(self, background_waves=0):
        """lazy mode, once per step AFTER the catch-up of the step's rows: replays slice (t-1 mod flush_every) of the table
        up to step t-1.  Rows of the current batch are already at t-1 and are skipped.  background_waves > 0: the capped,
        lowest-priority form of the launch (cdc_embed_lazy_flush_bg) for a second stream beside the forward/backward."""
        if self.table_mode != "lazy" or self.flush_every <= 1:
            return
        nb = 24.0 * self.table.numel() / self.flush_every / max(self.own_mod, 1)
        if background_waves > 0:
            L.launch("cdc_embed_lazy_flush(slice)", self.lib.cdc_embed_lazy_flush_bg,
                     (self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(),
                      self.table.shape[0], self.table.shape[1], self._hp(), self.step_dev.data_ptr(), -1, self.flush_every,
                      self.own_mod, self.own_rem, int(background_waves)), self._stream(), nbytes=nb)
            return
        L.launch("cdc_embed_lazy_flush(slice)", self.lib.cdc_embed_lazy_flush,
                 (self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(),
                  self.table.shape[0], self.table.shape[1], self._hp(), self.step_dev.data_ptr(), -1, self.flush_every,
                  self.own_mod, self.own_rem), self._stream(), nbytes=nb)

    def table_catchup_rows(self, idx, B, F, D, tag="", runs=0, flush=True):
        """lazy mode, BEFORE the gather of this step, on already computed row indices [B,F] (the gathered batch under DP).
        flush=False: the caller issues flush_slice() itself, later in the step."""
        assert self.table_mode == "lazy"
        s = self._stream()
        ws = self.sort_rows(idx, B, F, D, tag, runs)
        L.launch("cdc_embed_lazy_catchup", self.lib.cdc_embed_lazy_catchup,
                 (ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), self.table.data_ptr(), self.table_m.data_ptr(),
                  self.table_v.data_ptr(), self.table_last.data_ptr(), self._hp(), self.step_dev.data_ptr(), None, 0, B, F, D), s)
        if flush:
            self.flush_slice()

    def begin_step_sort(self, ids, offsets, B, F, D, tag="", err=None, begin=True):
        """begin_step() + table_index() + sort_rows() in the sort's launches: the sort reads the raw ids (row = id + field
        offset, out-of-range -> -1 and `err`), and its first launch also advances the step counter and clears the accumulators.
        begin=False: the sort alone (a batch sorted one step ahead: the step counter is not its business)."""
        ws = self._workspace(B, F, D, tag)
        L.launch("cdc_embed_sort_dedupe", self.lib.cdc_embed_sort_dedupe_ids,
                 (ids.data_ptr(), offsets.data_ptr(), self.table.shape[0], self.step_dev.data_ptr() if begin else None,
                  self.reg_sum.data_ptr() if begin else None, 2 if begin else 0,
                  None if err is None else err.data_ptr(),
                  ws["uniq"].data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), ws["cnt"].data_ptr(),
                  None if ws["scratch"] is None else ws["scratch"].data_ptr(), B, F), self._stream())
        return ws

    def catchup_sorted(self, B, F, D, tag=""):
        """lazy mode: the catch-up of the rows whose sorted list is in workspace `tag` (begin_step_sort(..., tag) of this or the
        previous step), without the slice (the caller issues flush_slice())."""
        assert self.table_mode == "lazy"
        ws = self._workspace(B, F, D, tag)
        L.launch("cdc_embed_lazy_catchup", self.lib.cdc_embed_lazy_catchup,
                 (ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), self.table.data_ptr(), self.table_m.data_ptr(),
                  self.table_v.data_ptr(), self.table_last.data_ptr(), self._hp(), self.step_dev.data_ptr(), None, 0, B, F, D), self._stream())

    def begin_step_catchup(self, ids, offsets, B, F, D, flush=True):
        """begin_step() + table_catchup() of a single-GPU lazy step with two launches less (begin_step_sort)."""
        assert self.table_mode == "lazy"
        s = self._stream()
        ws = self.begin_step_sort(ids, offsets, B, F, D)
        L.launch("cdc_embed_lazy_catchup", self.lib.cdc_embed_lazy_catchup,
                 (ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), self.table.data_ptr(), self.table_m.data_ptr(),
                  self.table_v.data_ptr(), self.table_last.data_ptr(), self._hp(), self.step_dev.data_ptr(), None, 0, B, F, D), s)
        if flush:
            self.flush_slice()

    def begin_step_catchup_gather(self, ids, offsets, out_ptr, out_h, ld_out_h, B, F, D, err=None, flush=True):
        """begin_step_catchup() whose catch-up launch also IS the gather of the step (cdc_embed_lazy_catchup_gather): the lanes
        that bring a row up to date write it to every batch position that looks it up — `out` [B, F*D] fp32 (+ the bf16 shadow)."""
        assert self.table_mode == "lazy"
        s = self._stream()
        ws = self.begin_step_sort(ids, offsets, B, F, D, err=err)
        L.launch("cdc_embed_lazy_catchup_gather", self.lib.cdc_embed_lazy_catchup_gather,
                 (ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), self.table.data_ptr(),
                  self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), self._hp(), self.step_dev.data_ptr(),
                  out_ptr, out_h, ld_out_h, B, F, D), s, nbytes=float(B) * F * (D * 4 + 4 + D * 4))
        if flush:
            self.flush_slice()

    def table_catchup(self, ids, offsets, idx, B, F, D, flush=True):
        """lazy mode, BEFORE the gather of this step: row indices -> dedupe -> replay the rows up to step t-1."""
        assert self.table_mode == "lazy"
        L.launch("cdc_embed_index", self.lib.cdc_embed_index,
                 (ids.data_ptr(), offsets.data_ptr(), idx.data_ptr(), None, B, F, self.table.shape[0]), self._stream())
        self.table_catchup_rows(idx, B, F, D, flush=flush)

    def table_step(self, idx, d_out, B, F, D, tag="", short_segments=False):
        """Adam step t on the table from the batch's row indices [B,F] and the gradient of the gathered rows [B,F*D].
        short_segments: no row occurs more than a few times in idx (an owner's received lists: once per sender)."""
        s = self._stream()
        hp = self._hp()
        w, m, v = self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr()
        if self.table_mode == "dense":
            ws = self.sort_rows(idx, B, F, D)
            self._segment_sum(ws, d_out, B, F, D, s)
            L.launch("cdc_embed_adam_touched", self.lib.cdc_embed_adam_touched,
                     (ws["rowgrad"].data_ptr(), ws["uniq"].data_ptr(), ws["cnt"].data_ptr(),
                      w, m, v, ws["side"].data_ptr(), hp, self.step_dev.data_ptr(), B, F, D), s)
            L.launch("cdc_embed_adam_dense_pass", self.lib.cdc_embed_adam_dense_pass,
                     (w, m, v, self.table.numel(), hp, self.step_dev.data_ptr(), self.reg_sum.data_ptr() + 8), s,
                     nbytes=24.0 * self.table.numel())
            L.launch("cdc_embed_adam_patch", self.lib.cdc_embed_adam_patch,
                     (ws["side"].data_ptr(), ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), w, m, v, B, F, D), s)
        else:
            ws = self._workspace(B, F, D, tag)   # rows were sorted by table_catchup of this step
            # per-row gradient sums and the rows' Adam step in one launch
            L.launch("cdc_embed_segsum_lazy_update", self.lib.cdc_embed_segsum_lazy_update,
                     (d_out.data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), ws["cnt"].data_ptr(), ws["uniq"].data_ptr(),
                      w, m, v, self.table_last.data_ptr(), hp, self.step_dev.data_ptr(), B, F, D, 1 if short_segments else 0), s)

    def _segment_sum(self, ws, d_out, B, F, D, s, short=False):
        """short: every segment is known to hold only a few entries (an owner's merged row lists) — no sorted copy"""
        if short:
            L.launch("cdc_embed_segment_sum", self.lib.cdc_embed_segment_sum_direct,
                     (d_out.data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), ws["cnt"].data_ptr(), ws["uniq"].data_ptr(),
                      ws["rowgrad"].data_ptr(), B, F, D), s)
            return
        L.launch("cdc_embed_segment_sum", self.lib.cdc_embed_segment_sum,
                 (d_out.data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), ws["cnt"].data_ptr(), None,
                  ws["rowgrad"].data_ptr(), B, F, D), s)

    def flush_table(self):
        """lazy mode: bring every row to the current step (needed before state_dict / eval / reading the table)."""
        if self.table_mode != "lazy":
            return
        L.launch("cdc_embed_lazy_flush", self.lib.cdc_embed_lazy_flush,
                 (self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(),
                  self.table.shape[0], self.table.shape[1], self._hp(), self.step_dev.data_ptr(), 0, 0, self.own_mod, self.own_rem),
                 self._stream(), nbytes=24.0 * self.table.numel())

    # ------------------------------------------------------------------------------------------
    def dense_step(self, param_grads, param_refs, slabs=None):
        """Adam step t on every dense parameter that has a gradient in the plan (params without one are skipped,
        as torch.optim.Adam skips `grad is None`).  slabs (plan.grad_slabs): gradients whose split-K slabs the grad-weight launches
        left unreduced — the launch adds them while it reads (address of the gradient tensor -> (first slab, stride, count))."""
        self._dense_prepare(param_grads, param_refs, slabs, False)
        s = self._stream()
        if getattr(self, "_dense_table", None) is not None:
            hdr, tab, wg_t, wg_c, n_wg = self._dense_table
            L.launch("cdc_adam_multi", self.lib.cdc_adam_multi_table, (C.byref(hdr), tab.data_ptr(), wg_t.data_ptr(), wg_c.data_ptr(), n_wg), s)
        for a in self._dense_args:
            L.launch("cdc_adam_multi", self.lib.cdc_adam_multi, (C.byref(a),), s)

    def rows_and_dense_step(self, idx, d_out, B, F, D, param_grads, param_refs, slabs=None, tag="", short_segments=False):
        """table_step (lazy table: per-row gradient sums + the rows' Adam step) and dense_step in ONE launch
        (cdc_embed_segsum_lazy_update_dense): what ends a single-GPU training step."""
        if self.table_mode != "lazy":
            raise RuntimeError("rows_and_dense_step: lazy table only")
        self._dense_prepare(param_grads, param_refs, slabs, True)
        if self._dense_table is None:            # (no dense parameter has a gradient)
            return self.table_step(idx, d_out, B, F, D, tag=tag, short_segments=short_segments)
        hdr, tab, wg_t, wg_c, n_wg = self._dense_table
        ws = self._workspace(B, F, D, tag)       # rows were sorted by the catch-up of this step
        L.launch("cdc_embed_segsum_lazy_update_dense", self.lib.cdc_embed_segsum_lazy_update_dense,
                 (d_out.data_ptr(), ws["seg"].data_ptr(), ws["perm"].data_ptr(), ws["cnt"].data_ptr(), ws["uniq"].data_ptr(),
                  self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), self._hp(),
                  self.step_dev.data_ptr(), B, F, D, 1 if short_segments else 0, C.byref(hdr), tab.data_ptr(), wg_t.data_ptr(), wg_c.data_ptr(),
                  n_wg), self._stream())

    def _dense_prepare(self, param_grads, param_refs, slabs, table_form):
        """argument blocks of the dense Adam launch(es) for this gradient layout -> self._dense_args / self._dense_table.
        table_form: descriptors and workgroup map as device arrays whatever the tensor count."""
        slabs = slabs or {}
        sig = (tuple((k, g.data_ptr(), slabs.get(g.data_ptr())) for k, g in param_grads.items()), bool(table_form))
        # one argument set per gradient layout (= per plan stepping this optimiser: a training step and its sibling for the ragged
        # batch alternate), ALL kept alive: a captured graph holds the address of the device-resident descriptor table it was
        # captured with (round 3 kept only the last one: a sibling's step freed the table the main step's graphs point to)
        cache = self.__dict__.setdefault("_dense_cache", {})
        hit = cache.get(sig)
        if hit is not None:
            self._dense_args, self._dense_table = hit
            self._dense_sig = sig
        if self._dense_sig != sig:
            items = []
            for k, g in param_grads.items():
                p = param_refs[k]
                if p is self.table or id(p) in self.frozen:
                    continue
                st = self.state.get(k)
                if st is None:
                    st = (torch.zeros_like(p.data), torch.zeros_like(p.data))
                    self.state[k] = st
                items.append((p, g, st))
            self._dense_table = None
            if len(items) > L.MAX_TENSORS or (table_form and items):
                # more tensors than one kernel-argument block holds: ONE launch whose descriptors and workgroup map are device arrays
                tab = (L.AdamTensor * len(items))()
                wg_t, wg_c = [], []
                for i, (p, g, st) in enumerate(items):
                    T = tab[i]
                    T.w, T.g, T.m, T.v, T.n = p.data_ptr(), g.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), p.numel()
                    T.l2 = float(torch.tensor(self._l2_of.get(id(p), 0.0), dtype=torch.float64).to(torch.float32))
                    sl = slabs.get(g.data_ptr())
                    T.slabs, T.slab_stride, T.n_slabs = sl if sl is not None else (None, 0, 0)
                    nck = -(-p.numel() // L.ADAM_CHUNK)
                    wg_t += [i] * nck
                    wg_c += list(range(nck))
                hdr = L.AdamArgs()
                hdr.n_tensors = 0
                hdr.lerp_w, hdr.beta2, hdr.one_minus_beta2, hdr.eps, hdr.weight_decay = self._lerp_w, self._beta2, self._omb2, self._eps, self._wd
                hdr.step_scalars, hdr.n_scalars = self.scalars.data_ptr(), self.scalars.shape[0]
                hdr.grad_scale = self.grad_scale
                hdr.step_dev = self.step_dev.data_ptr()
                hdr.reg_sum = self.reg_sum.data_ptr()
                hdr.reg_seed = self.table_reg.data_ptr() if self.table_mode == "lazy" else 0
                dev_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
                self._dense_table = (hdr, dev_tab, torch.tensor(wg_t, dtype=torch.int32, device=self.device),
                                     torch.tensor(wg_c, dtype=torch.int32, device=self.device), len(wg_t))
                items = []
            args = []
            for c0 in range(0, len(items), L.MAX_TENSORS):
                a = L.AdamArgs()
                chunk = items[c0:c0 + L.MAX_TENSORS]
                a.n_tensors = len(chunk)
                a.lerp_w, a.beta2, a.one_minus_beta2, a.eps, a.weight_decay = self._lerp_w, self._beta2, self._omb2, self._eps, self._wd
                a.step_scalars, a.n_scalars = self.scalars.data_ptr(), self.scalars.shape[0]
                a.grad_scale = self.grad_scale
                a.step_dev = self.step_dev.data_ptr()
                a.reg_sum = self.reg_sum.data_ptr()
                # lazy table: the step's reg figure = dense parameters' sum + the table term cached by refresh_table_reg()
                a.reg_seed = self.table_reg.data_ptr() if (self.table_mode == "lazy" and c0 == 0) else 0
                for i, (p, g, st) in enumerate(chunk):
                    T = a.t[i]
                    T.w, T.g, T.m, T.v, T.n = p.data_ptr(), g.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), p.numel()
                    T.l2 = float(torch.tensor(self._l2_of.get(id(p), 0.0), dtype=torch.float64).to(torch.float32))
                    sl = slabs.get(g.data_ptr())
                    if sl is not None:
                        T.slabs, T.slab_stride, T.n_slabs = sl
                    else:
                        T.slabs, T.slab_stride, T.n_slabs = None, 0, 0
                args.append(a)
            self._dense_args, self._dense_sig = args, sig
            cache[sig] = (args, self._dense_table)

    def reg_loss(self):
        """device double: the step's regularisation term sum(l2*w^2) (dense params + table).  Lazy mode: the table's part is the
        value cached by refresh_table_reg(), already added by the dense step (cdc_adam_args.reg_seed)."""
        if self.table_mode == "lazy":
            return self.reg_sum[0]
        return self.reg_sum[0] + self.l2_table * self.reg_sum[1]

    def table_reg_loss(self, owned_only=False):
        """l2 * sum(w^2) over the table after bringing its rows to the current step (model/layer.py:31,96-112).  owned_only (the
        row-sharded table under data parallelism): only the rows this rank maintains (r % own_mod == own_rem) — the others are
        stale here; the caller adds the ranks' figures up.  Summed in double over row blocks (no table-sized temporary)."""
        self.flush_table()
        w = self.table.data
        if owned_only and self.own_mod > 1:
            w = w[self.own_rem::self.own_mod]
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        step = 1 << 22
        for r0 in range(0, w.shape[0], step):
            total += torch.sum(torch.square(w[r0:r0 + step].double()))
        return self.l2_table * total

    def refresh_table_reg(self, dist=None):
        """Lazy table: re-evaluates the table's share of the reference's reported loss (run.py:489) for the weights the next
        forward sees and caches it in `table_reg`, which every step adds to its `reg` figure until the next refresh.  dist: the
        data-parallel group when the table is row-sharded (owned rows summed per rank, then added up across ranks)."""
        if self.table_mode != "lazy":
            return self.table_reg
        sharded = dist is not None and self.own_mod > 1
        val = self.table_reg_loss(owned_only=sharded)
        if sharded:
            val = val.reshape(1).clone()
            dist.all_reduce_sum(val)
            val = val[0]
        self.table_reg.copy_(val)
        self.table_reg_ready = True
        return self.table_reg

    # ---- checkpointing (run.py:447: 'optimizer': optimizer.state_dict()) -------------------------------------
    def state_dict(self):
        self.flush_table()
        names = {id(p): n for n, p in self.model.named_parameters()}
        st = {names[k]: {"exp_avg": m.clone(), "exp_avg_sq": v.clone()} for k, (m, v) in self.state.items() if k in names}
        st[names[id(self.table)]] = {"exp_avg": self.table_m.clone(), "exp_avg_sq": self.table_v.clone()}
        return {"step": int(self.step_dev.item()), "state": st,
                "hyper": {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}}

    def load_state_dict(self, sd):
        params = dict(self.model.named_parameters())
        self.step_dev.fill_(int(sd["step"]))
        for n, st in sd["state"].items():
            p = params[n]
            if p is self.table:
                self.table_m.copy_(st["exp_avg"])
                self.table_v.copy_(st["exp_avg_sq"])
            else:
                # cdc_adam_multi's argument blocks (and any hipGraph captured from them) hold the moment POINTERS: restore into
                # the tensors that already exist, allocate only what the optimiser has never seen
                have = self.state.get(id(p))
                if have is not None and have[0].shape == st["exp_avg"].shape:
                    have[0].copy_(st["exp_avg"])
                    have[1].copy_(st["exp_avg_sq"])
                else:
                    self.state[id(p)] = (st["exp_avg"].to(self.device).clone(), st["exp_avg_sq"].to(self.device).clone())
                    self._dense_sig = None
                    self.__dict__.pop("_dense_cache", None)
        if self.table_last is not None:
            self.table_last.fill_(int(sd["step"]))
        self.table_reg_ready = False
