"""Step driver of the HIP path: one training step with the reference's order of operations
(Run.train, run.py:470-497):

    pred = model(X) ; loss = BCE(pred.gather(1, group), y) ; loss += model.get_regularization_loss()
    model.zero_grad() ; loss.backward() ; optimizer.step() ; loss.item()

Here the whole step is a fixed sequence of launches (plan forward -> BCE+grad -> plan backward -> table and
dense Adam with the L2 term folded in), replayable as ONE hipGraph.  Nothing synchronises with the host:
`TrainStep.step()` returns device scalars; read them (loss.item()) only when you log.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from .optim import FusedAdam


class TrainStep:
    """mode:
         "multi"   multi-tower models (PLE / MMoE / CDC split): pred = model(X).gather(1, group)  (run.py:481-484)
         "single"  DCN / DCNv2: pred = model(X)                                                   (run.py:486-488)
         "mean"    CDC warm-up: pred = mean over the towers of model(X)                           (cdc.py:100-102, run.py:616)
         "single_group"  one output column, but the model itself reads the row's group (HiNet: model(X, group))  (run.py:476-479)
         "star"    STAR: pred, y = model(X, group, targets=y) — rows partitioned by domain, every   (run.py:476-479,
                   partition through its own tower, loss on the group-ordered targets              star.py:109-181)
       train_mode=False: the step runs with the module in EVAL mode (BatchNorm on its running statistics, no dropout) but
       still back-propagates and updates — what the reference's CDC loop does after its first evaluation pass, which
       leaves the model in eval() (run.py:550-551 sets it, nothing sets it back before the training that follows)."""

    def __init__(self, model, optimizer: FusedAdam, batch_size, mode="multi", use_graph=False, dist=None, sync_bn=True,
                 table_dist=None, shard_slack=1.5, train_mode=True, overlap=True, overlap_waves=2, sort_ahead=True, fuse_gather=False,
                 defer_dw_reduce=None, global_rows=None, cap_rows=None, tower_one_launch=True, rows_dense_one_launch=True):
        """sync_bn (data parallel only): BatchNorm statistics over the GLOBAL batch, as the reference's single process
        computes them — two small all-reduces per BatchNorm launch; False = per-rank statistics.
        table_dist (data parallel only): "sharded" (row r owned by rank r % world; default with the lazy table optimiser)
        or "replicated" (every rank applies the global batch's table update).  shard_slack: capacity of the per-owner,
        per-field row lists as a multiple of the even share B/world (an overflow raises in check_ids()).
        overlap / overlap_waves / sort_ahead / fuse_gather (single GPU, lazy table; for tests and A/B runs — the defaults are what is
        global_rows / cap_rows (data parallel, the ragged last global batch of an epoch: TrainStep.sibling): the batch's true global
        size (the BCE mean runs over it; this rank's share is batch_size) and the largest share any rank holds (row-list capacities
        are a property of the exchange, equal on every rank).
        measured and shipped): the replay slice in the background on a second chain (waves per SIMD of its capped grid), the next
        batch's row sort on that chain (step(..., next_X=)), the catch-up launch that also writes the gathered embeddings (slower:
        profiles/round3/README.md section 3)."""
        self.model, self.opt, self.B, self.mode = model, optimizer, int(batch_size), mode
        self.lib = L.load()
        self.dist = dist
        self.sync_bn = bool(sync_bn)
        self.world = 1 if dist is None else dist.world_size
        self.dp_on = self.world > 1 or bool(getattr(dist, "force", False))     # force: the collective path with one rank
        self.global_B = self.B * self.world if global_rows is None else int(global_rows)
        self._shard_slack = float(shard_slack)
        dev = optimizer.device
        self.device = dev
        self.train_mode = bool(train_mode)
        self._overlap_ok, self._overlap_waves = bool(overlap), int(overlap_waves)
        self._sort_ahead_wanted, self._fuse_gather_wanted = sort_ahead, bool(fuse_gather)
        # the two updates that end a single-GPU step with the lazy table (the step's rows; the dense parameters) as one launch
        self._rows_dense_one = bool(rows_dense_one_launch) and not self.dp_on
        self.tower_one_launch = bool(tower_one_launch)        # the fused towers' forward and backward as one launch (cdc_tower_step)
        # split-K slabs of the batched grad-weight launches summed by the dense Adam launch (single GPU; under data parallelism the
        # reduced gradient is what is all-reduced)
        self._defer_dw = (not self.dp_on) if defer_dw_reduce is None else (bool(defer_dw_reduce) and not self.dp_on)
        assert mode in ("multi", "single", "mean", "star", "single_group")
        # the plan shares the optimiser's step counter (dropout stream) and its flat gradient arena
        self.holder = self._build_plan()
        self.plan = self.holder.plan
        self.emb = self.holder.emb_op
        self.out = self.holder.outputs[0]
        self.y = torch.zeros(self.B, dtype=torch.int16, device=dev)
        self.group = torch.zeros(self.B, dtype=torch.int64, device=dev) if mode == "multi" else None
        if mode == "single_group":
            self.group_in = self.holder.inputs[1]                 # int64 [B] read by the plan's group-selection launch
        if mode == "star":
            self.group = self.holder.inputs[1]                    # the partition kernel reads the domain of every row here
            self.order = self.holder.extra_outputs[0]             # row order after the partition (ascending group)
            self.y_perm = torch.zeros(self.B, dtype=torch.int16, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        # ids are checked against their FIELD's vocabulary while a batch is staged (an id past it that still lies inside the table
        # aliases another field's row: check_ids reports it)
        self.field_dims_dev = torch.tensor([int(d) for d in model.feature_dims], dtype=torch.int32, device=dev)
        self.alias = torch.zeros(1, dtype=torch.int32, device=dev)
        # the step's sum(l2*w^2), formed inside the launch sequence; lazy table: the optimiser's accumulator itself (see _reg)
        self.reg = self.opt.reg_sum[0] if self.opt.table_mode == "lazy" else torch.zeros((), dtype=torch.float64, device=dev)
        self.use_graph = use_graph
        self.graph = None
        self._stage_graphs = {}
        self._step_graphs = {}            # data parallel over RCCL: the whole step (launches + collectives) as one graph per sequence
        self._one_graph_ok = None
        self._dp_seq = None
        self._warm = 0
        self._reg_checked = False
        if table_dist is None:
            table_dist = "sharded" if optimizer.table_mode == "lazy" else "replicated"
        assert table_dist in ("sharded", "replicated")
        self.table_dist = table_dist if self.dp_on else "replicated"
        if self.dp_on and self.table_dist == "replicated":
            F, D = self.emb.F, self.emb.D
            self.idx_all = torch.empty((self.global_B, F), dtype=torch.int32, device=dev)
            self.dE_all = torch.empty((self.global_B, F * D), dtype=torch.float32, device=dev)
            self._loss_behind_arena()
        if self.dp_on and self.table_dist == "sharded":
            if optimizer.table_mode != "lazy":
                raise ValueError("the row-sharded table needs table_mode='lazy'")
            F, D, N = self.emb.F, self.emb.D, self.world
            Bc = self.B if cap_rows is None else int(cap_rows)           # the largest local batch of the step over the ranks
            assert Bc >= self.B
            cap = min(Bc, int(-(-Bc // N) * float(shard_slack)) + 16)
            if N * cap > L.SORT_MAX_ROWS:
                raise ValueError(f"world {N} x list capacity {cap} exceeds the per-field sort limit {L.SORT_MAX_ROWS}")
            self.cap, self.Bv = cap, N * cap                          # Bv: rows of the owner-side batch
            z = dict(device=dev)
            self.send_ids = torch.full((N, cap, F), -1, dtype=torch.int32, **z)
            self.recv_ids = torch.full((N, cap, F), -1, dtype=torch.int32, **z)
            self.recv_ids_alt = torch.full((N, cap, F), -1, dtype=torch.int32, **z)   # the id exchange one step ahead lands here
            self.slot_of = torch.zeros((F, self.B), dtype=torch.int32, **z)
            self.overflow = torch.zeros(1, dtype=torch.int32, **z)
            self.rows_send = torch.zeros((N, cap, F * D), dtype=torch.float32, **z)
            self.rows_recv = torch.zeros((N, cap, F * D), dtype=torch.float32, **z)
            self.grads_send = torch.zeros((N, cap, F * D), dtype=torch.float32, **z)
            self.grads_recv = torch.zeros((N, cap, F * D), dtype=torch.float32, **z)
            self.zero_offsets = torch.zeros(F, dtype=torch.int32, **z)
            optimizer.own_mod, optimizer.own_rem = N, dist.rank
            self._loss_behind_arena()

    def _loss_behind_arena(self):
        """Data parallel: the loss scalar sits right behind the used part of the flat gradient arena, so ONE all-reduce covers the
        dense gradients and the loss."""
        used = max(self.plan._arena_used, 1)
        assert self.opt.grad_arena.numel() > used
        self.arena_and_loss = self.opt.grad_arena[:used + 1]
        self.loss = self.opt.grad_arena[used:used + 1]

    def _build_plan(self):
        model, opt = self.model, self.opt
        B = self.B
        dev = opt.device
        from .functional import PlanHolder
        from . import plan as P

        def build():
            plan = P.Plan(dev, B, precision=model.precision, training=self.train_mode,
                          dropout=float(getattr(model, "dropout_p", 0.0)) if self.train_mode else 0.0,
                          seed=int(getattr(model, "seed", 0)), step_dev=opt.step_dev, grad_arena=opt.grad_arena,
                          dist=self.dist if self.sync_bn else None, defer_dw_reduce=self._defer_dw)
            emb = model.embedding.describe(plan)
            outs, ins, extra = model.describe(plan, emb, grouped=True) if self.mode == "star" else model.describe(plan, emb)
            plan.finalize(outs)
            return PlanHolder(plan, [emb.ids] + ins, outs, emb_op=emb, extra_outputs=extra)

        was = model.training
        model.train(self.train_mode)                 # the ops read module.training while describing themselves
        try:
            return model._cache().get(model, ("train_step", id(opt), self.train_mode, self.mode == "star",
                                               bool(self.sync_bn and self.dp_on), bool(self.dp_on), self._defer_dw), B, build)
        finally:
            model.train(was)

    # ------------------------------------------------------------------------------------------
    def _find_head(self):
        """The launch that turns the tower logits into the predictions the loss reads: a single sigmoid row-dot launch whose
        groups are the columns of `self.out`, in order.  Its backward can form the BCE gradient itself (cdc_rowdot_bwd's fused
        BCE): one launch and a round trip through d_out less per step.  Returns that launch's argument block or None."""
        from . import plan as P
        out = self.out
        for op in self.plan.ops:
            if isinstance(op, (P.TowerHead, P.TowerChain)) and op.sigmoid and op.out.root is out.root and op.out.col0 == out.col0 and op.out.cols == out.cols:
                args = getattr(op, "bwd_args", [])
                if len(args) == 1 and op.M == self.B:
                    self._head_op = op
                    return args[0]
                return None
        for op in self.plan.ops:
            if not isinstance(op, P.RowDot) or not op.sigmoid or op.row_offsets is not None or len(op.groups) != out.cols:
                continue
            if all(g["out"].root is out.root and g["out"].col0 == out.col0 + i and g["out"].cols == 1 for i, g in enumerate(op.groups)):
                args = getattr(op, "bwd_args", [])
                return args[0] if len(args) == 1 and op.M == self.B else None
        return None

    def _head_lookup(self):
        head = self.__dict__.get("_head", False)
        if head is False:
            self._head_op = None
            head = self._head = self._find_head()
            self._fuse_bce = head is not None and self.mode in ("multi", "single", "single_group")
            if self._fuse_bce:
                self._bce_partial = torch.zeros(L.MAX_GROUPS * L.ROWDOT_PARTS, dtype=torch.float64, device=self.device)
            from . import plan as P
            # the towers' two launches as one (csrc/tower.hip cdc_tower_step): the fused loss, one GPU
            self._tower_both = (self._head_op if self._fuse_bce and self.tower_one_launch and not self.dp_on and
                                isinstance(self._head_op, P.TowerChain) and self.plan.dist is None else None)
        return head

    def _towers_one_launch(self, on):
        """around the launches of a step (the plan, and with it the op, may be shared with a trainer that wants the other form)"""
        self._head_lookup()
        if self._tower_both is not None:
            self._tower_both.one_launch = bool(on)

    def _bce(self):
        og = self.out.grad
        head = self._head_lookup()
        if head is not None:
            # the argument block belongs to the (cached, possibly shared) plan: set for this trainer before every backward
            if self._fuse_bce:
                head.bce_group = None if self.group is None else self.group.data_ptr()
                head.bce_y_i16, head.bce_y_f32 = self.y.data_ptr(), None
                head.bce_loss = self.loss.data_ptr()
                if hasattr(type(head), "bce_partial"):            # (the fused tower launch keeps its partial sums in its own workspace)
                    head.bce_partial = self._bce_partial.data_ptr()
                head.bce_inv_count = 1.0 / self.global_B
                return
            head.bce_y_i16 = head.bce_y_f32 = None
        if self.mode == "mean":
            L.launch("cdc_bce_mean_fwd_bwd", self.lib.cdc_bce_mean_fwd_bwd,
                     (self.out.ptr, self.out.ld, self.y.data_ptr(), None, self.loss.data_ptr(), og.ptr, og.ld, self.B, self.out.cols,
                      1.0 / self.global_B), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            return
        y, group = self.y, self.group
        if self.mode == "star":                                   # targets[order] (star.py:181), one column
            torch.index_select(self.y, 0, self.order.long(), out=self.y_perm)
            y, group = self.y_perm, None
        L.launch("cdc_bce_fwd_bwd", self.lib.cdc_bce_fwd_bwd,
                 (self.out.ptr, self.out.ld, None if group is None else group.data_ptr(), y.data_ptr(), None,
                  self.loss.data_ptr(), og.ptr, og.ld, self.B, self.out.cols, 1.0 / self.global_B),
                 C.c_void_p(torch.cuda.current_stream().cuda_stream))

    def _fuse_gather(self):
        """single GPU, lazy table: the catch-up launch writes the gathered embeddings itself (csrc/embedding.hip
        k_lazy_catchup_gather); needs a row's 16-byte lanes inside one wave"""
        hit = self.__dict__.get("_fuse_gather_ok")
        if hit is None:
            import os
            D = self.emb.D
            # off by default: measured at C2 (profiles/round2/README.md) the fused launch takes 48 us against 28.5 + 7.2 us for
            # catch-up + gather — the wave that owns the domain column's three rows writes ~B positions on its own while the rest
            # of the chip has finished; it wins only where no row is looked up by more than a few samples
            hit = D % 4 == 0 and 64 % (D // 4) == 0 and self._fuse_gather_wanted
            self._fuse_gather_ok = hit
            self._fwd_after_gather = [s_ for s_ in self.plan.fwd_steps if s_ is not self.emb.fwd_step]
        return hit

    def _reg(self):
        # part of the (graph-replayed) launch sequence, so that step() issues nothing else per call.  Dense table mode: the
        # streaming pass sums w^2 of the whole table every step (reg_sum[1]).  Lazy table: nobody walks the table in a step;
        # the table's term is the value of the last refresh_table_reg() (flush + exact sum), see there; the first dense Adam launch
        # of the step adds it to the dense parameters' sum, so `reg` IS the optimiser's accumulator and no launch is issued here.
        if self.opt.table_mode != "lazy":
            torch.add(self.opt.reg_sum[0], self.opt.reg_sum[1], alpha=self.opt.l2_table, out=self.reg)

    def refresh_table_reg(self):
        """Lazy table: brings every row to the current step and re-evaluates the table's share of the reference's reported loss
        (run.py:489: `loss += get_regularization_loss()`, i.e. l2 * sum(w^2) over the WHOLE table, model/layer.py:31,96-112).  The
        value is exact for the weights the NEXT step's forward sees, and is what `step()` adds to its reg figure until the next
        refresh (per step it changes by ~lr * 2 * l2 * sum|w|: 2e-8 relative at the reference's settings).  One pass over
        the table: called when the step is built, at the start of an epoch and where the reference's value is looked at (the
        logging steps of train_epoch / CDCTrainer), not every step.  Row-sharded table: every rank sums the rows it owns and the
        figures are added up across ranks (a collective: call it on every rank)."""
        sharded = self.dp_on and self.table_dist == "sharded"
        return self.opt.refresh_table_reg(self.dist if sharded else None)

    def _launch_all(self):
        self._towers_one_launch(True)
        try:
            self._launch_all_seq()
        finally:
            self._towers_one_launch(False)

    def _launch_all_seq(self):
        """single GPU: the whole step is one launch sequence (one hipGraph when use_graph)."""
        opt, plan, emb = self.opt, self.plan, self.emb
        B, F, D = self.B, emb.F, emb.D
        if opt.table_mode == "lazy":
            # sort -> catch-up of the batch's rows [which also writes the gathered embeddings (+ shadow): hot rows staged in LDS and
            # written by a whole workgroup, the plan's own gather launch skipped] -> this step's slice of the whole-table replay.
            # With the slice in the BACKGROUND, two chains on two hardware queues (two branches of the hipGraph when captured):
            #   main:  sort, catch-up(+gather) | forward, BCE, backward, grad-weight launches, dense Adam | per-row sums + Adam on the
            #   side:                          | the slice                                                |   step's rows, reg
            # The slice is VALU-only and touches no row of the batch (those are at step t-1 after the catch-up and are skipped); it
            # goes out capped (two waves per SIMD) at the lowest issue priority (cdc_embed_lazy_flush_bg) while the chain's kernels
            # raise theirs (CDC_PRIO_MAIN): it takes the issue cycles the chain leaves idle while waiting on L2 / LDS / MFMA results.
            # The table update needs the slice finished: it goes last on the main chain, behind the one join.
            fused, bg = self._fuse_gather(), self._overlap()
            if fused:
                oh, ldh = (plan.shadow_view(emb.out) if (plan.use_g2 and plan.has_shadow(emb.out)) else (None, 0))
                opt.begin_step_catchup_gather(emb.ids, emb.offsets, emb.out.ptr, oh, ldh, B, F, D, err=emb.err, flush=not bg)
            else:
                opt.begin_step_catchup(emb.ids, emb.offsets, B, F, D, flush=not bg)
            main = torch.cuda.current_stream()
            if bg:
                side = self._side_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    opt.flush_slice(background_waves=self._overlap_waves)
            if fused:
                st = C.c_void_p(main.cuda_stream)
                for fn in self._fwd_after_gather:
                    fn(st)
            else:
                plan.forward()
            self._bce()
            plan.backward()
            if self._rows_dense_one:
                if bg:
                    main.wait_stream(side)
                opt.rows_and_dense_step(emb.idx, emb.out.grad.root, B, F, D, plan.param_grads, plan._param_refs, plan.grad_slabs)
                self._reg()
                return
            opt.dense_step(plan.param_grads, plan._param_refs, plan.grad_slabs)
            if bg:
                main.wait_stream(side)
            opt.table_step(emb.idx, emb.out.grad.root, B, F, D)
            self._reg()
            return
        opt.begin_step()
        plan.forward()
        self._bce()
        plan.backward()
        opt.table_step(emb.idx, emb.out.grad.root, B, F, D)
        opt.dense_step(plan.param_grads, plan._param_refs, plan.grad_slabs)
        self._reg()

    def _sort_ahead(self):
        """Single GPU, lazy table, slice in the background: the sort + dedupe of batch t+1's rows (30 us of three dependent launches
        that read nothing but the ids) goes out on the side chain of step t, behind the slice, when the caller names the next batch
        (step(..., next_X=)); step t+1 then starts at the catch-up.  TrainStep(sort_ahead=False): every step sorts its own batch first."""
        if getattr(self, "_ahead_ok", None) is None:
            self._ahead_ok = (not self.dp_on and self.opt.table_mode == "lazy" and self._overlap() and not self._fuse_gather() and
                              bool(self._sort_ahead_wanted))
            if self._ahead_ok:
                self.ids_next = torch.zeros_like(self.emb.ids)
                self._parity, self._sorted_for, self._graphs = 0, None, {}
        return self._ahead_ok

    def _launch_ahead(self, have, prefetch, p):
        self._towers_one_launch(True)
        try:
            self._launch_ahead_seq(have, prefetch, p)
        finally:
            self._towers_one_launch(False)

    def _launch_ahead_seq(self, have, prefetch, p):
        """_launch_all's lazy branch with the row sort out of the chain and the catch-up on the SIDE chain: have = this batch's
        sorted rows are in workspace a<p> (sorted by the previous step), prefetch = sort the batch in ids_next into workspace
        a<1-p>.  begin_step's work was done by the staging launch (cdc_stage_batch_next).
            side:  [sort] catch-up * | slice, [sort of the next batch]
            main:  weight shadows    | (wait *) gather, forward, BCE, backward, grad-weight launches, dense Adam | join | row update
        The fork sits at the head of the step: the weight-shadow launch reads nothing of the table and covers the cross-queue edge
        (by the kernel trace, profiles/round3/step_timeline.txt, the fork-after-catch-up form left 12 us between the catch-up and
        the first forward launch)."""
        opt, plan, emb = self.opt, self.plan, self.emb
        B, F, D = self.B, emb.F, emb.D
        cur, nxt = f"a{p}", f"a{1 - p}"
        main = torch.cuda.current_stream()
        side = self._side_stream()
        # measured (same box, ms/step): fork after the catch-up + row update behind the join 0.411 / 0.412; catch-up on the side chain
        # under the weight-shadow launch 0.403; row update on the side chain as well (waiting for dE by an event, beside the
        # grad-weight launches and the dense Adam) 0.420 / 0.424 — two more cross-queue edges cost more than the 24 us they free;
        # main chain captured FIRST behind the catch-up (so that it stays on the origin queue and the side chain takes the
        # cross-queue edge): the replayed graph then starts the slice 150 us late, 0.476
        st = C.c_void_p(main.cuda_stream)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if not have:
                opt.begin_step_sort(emb.ids, emb.offsets, B, F, D, tag=cur, begin=False)
            opt.catchup_sorted(B, F, D, tag=cur)
            rows_ready = torch.cuda.Event()
            rows_ready.record(side)
            opt.flush_slice(background_waves=self._overlap_waves)
            if prefetch:          # behind the slice (ahead of it the sort delays the whole side chain into the backward: 0.401 -> 0.443 ms;
                                  # held back by an event until the backward chain has run: 0.408 -> 0.427)
                opt.begin_step_sort(self.ids_next, emb.offsets, B, F, D, tag=nxt, begin=False)
        nws = getattr(plan, "n_wshadow_steps", 0)
        for fn in plan.fwd_steps[:nws]:
            fn(st)
        main.wait_event(rows_ready)
        for fn in plan.fwd_steps[nws:]:
            fn(st)
        self._bce()
        late = set(id(s_) for s_ in plan.deferred_dw_steps)
        for fn in plan.bwd_steps:
            if id(fn) not in late:
                fn(st)
        for fn in plan.deferred_dw_steps:
            fn(st)
        if self._rows_dense_one:
            # the step's rows and the dense parameters in one launch behind the join (optim.rows_and_dense_step)
            main.wait_stream(side)
            opt.rows_and_dense_step(emb.idx, emb.out.grad.root, B, F, D, plan.param_grads, plan._param_refs, plan.grad_slabs, tag=cur)
            self._reg()
            return
        opt.dense_step(plan.param_grads, plan._param_refs, plan.grad_slabs)
        # (the row update on the side chain beside these two, or on this chain in front of them as soon as the slice has ended, was
        # measured again in round 4: 0.345 and 0.350 ms against 0.338 — every additional cross-queue edge of the replayed graph moves
        # nodes to another hardware queue and costs ~10 us where it lands, profiles/round4/README.md section 6)
        main.wait_stream(side)
        opt.table_step(emb.idx, emb.out.grad.root, B, F, D, tag=cur)
        self._reg()

    def _step_ahead(self, X, nx):
        # the remembered tensor is kept alive (its storage cannot be handed to another batch) and its version pins its contents
        key = (X.data_ptr(), X._version) if torch.is_tensor(X) else None
        have = self._sorted_for is not None and self._sorted_for[:2] == key
        prefetch = nx is not None
        p = self._parity
        if self.use_graph and self._warm >= 2:
            g = self._graphs.get((have, prefetch, p))
            if g is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch_ahead(have, prefetch, p)
                self._graphs[(have, prefetch, p)] = g
                self.graph = g
            g.replay()
        else:
            self._launch_ahead(have, prefetch, p)
            self._warm += 1
        if prefetch:
            self._parity = 1 - p
            self._sorted_for = (nx.data_ptr(), nx._version, nx)
        else:
            self._sorted_for = None

    def _overlap(self):
        """The replay slice of the lazy table on a second stream beside the forward/backward (single GPU).  TrainStep(overlap=False)
        puts it back on the main chain, overlap_waves sets the slice's waves per SIMD.  Measured at C2 (profiles/round3/README.md):
        serial 0.550 ms/step; side by side with the full grid 0.596 (round 2: its 4096 workgroups take every wave slot); capped at
        two waves per SIMD, lowest priority, table update behind a single join: 0.515."""
        return self._overlap_ok

    def _drop_graphs(self):
        """releases every captured graph (before the process group is destroyed: DataParallel.close)"""
        self._step_graphs.clear()
        self._stage_graphs.clear()
        self.graph = None
        if hasattr(self, "_graphs"):
            self._graphs.clear()

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # ---- data parallel: launch segments separated by the collectives (each segment replayable as a graph) ---------------
    def _dp_sequence(self):
        """[(is_comm, fn)]: dense gradients: ONE sum all-reduce of the flat arena (the loss already carries
        1/global_batch); table: every rank applies the identical update from the all-gathered (row index, row gradient)
        pairs of the global batch; BatchNorm: statistics all-reduced inside the plan (sync_bn)."""
        opt, plan, emb, dp = self.opt, self.plan, self.emb, self.dist
        F, D = emb.F, emb.D

        def stage0():
            opt.begin_step()
            opt.table_index(emb.ids, emb.offsets, emb.idx, self.B, F)

        def catchup():
            if opt.table_mode == "lazy":
                opt.table_catchup_rows(self.idx_all, self.global_B, F, D)

        def exchange():
            dp.all_reduce_sum(self.arena_and_loss)                      # dense gradients + the loss scalar behind them
            dp.all_gather_rows(self.dE_all, emb.out.grad.root)

        def update():
            opt.table_step(self.idx_all, self.dE_all, self.global_B, F, D)
            opt.dense_step(plan.param_grads, plan._param_refs, plan.grad_slabs)
            self._reg()

        def run_steps(steps):
            def fn():
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for s_ in steps:
                    s_(st)
            return fn

        seq = [(False, stage0), (True, lambda: dp.all_gather_rows(self.idx_all, emb.idx)), (False, catchup)]
        for is_comm, steps in plan.segments(plan.fwd_steps):
            seq.append((is_comm, run_steps(steps)))
        seq.append((False, self._bce))
        for is_comm, steps in plan.segments(plan.bwd_steps):
            seq.append((is_comm, run_steps(steps)))
        seq += [(True, exchange), (False, update)]
        merged = []                                   # fuse neighbouring launch pieces into one segment
        for is_comm, fn in seq:
            if merged and not is_comm and not merged[-1][0]:
                merged[-1][1].append(fn)
            else:
                merged.append((is_comm, [fn]))
        return merged

    def _dp_sequence_sharded(self, ahead=False, have=False, prefetch=False, p=0):
        """Row-sharded table: [(is_comm, [fn...])].  Per step and rank the table work is that of the LOCAL batch:
            sort local rows -> bucket by owner | a2a ids | owner: merge the sorted lists, catch-up, gather |
            a2a rows | expand, forward, BCE, backward, per-row gradient sums, pack | all-reduce arena, a2a grads |
            owner: per-row sums over the senders, Adam update of its rows; dense Adam.
        ahead (step(..., next_X=) in use): the step counter is advanced by the staging launch, and the first two items — which
        read nothing but ids — run one step AHEAD: with `prefetch` the NEXT batch (ids_next) is sorted and bucketed in the segment
        of the grad-weight launches (the local sort workspace and the slot lists are free once `pack` has run) and its id exchange
        is started behind the gradient all-reduce, into the OTHER of two receive buffers (this step's owner update still reads
        its own); with `have` this step starts at the owner's `serve`, after awaiting that exchange.  p: which receive buffer
        holds THIS step's lists."""
        opt, plan, emb, dp = self.opt, self.plan, self.emb, self.dist
        B, F, D, N, cap, Bv = self.B, emb.F, emb.D, self.world, self.cap, self.Bv
        lib = self.lib
        bufs = (self.recv_ids, self.recv_ids_alt)
        recv_cur, recv_nxt = bufs[p], bufs[1 - p]
        # round 4: with the whole step in ONE graph (or issued eagerly) a second chain can run beside the dense one, as on one GPU:
        # the owner's replay slice (capped, lowest priority) from the catch-up to the row update, and the next batch's sort and
        # bucketing beside the grad-weight launches.  Launch SEGMENTS cannot hold a fork in one and its join in another: there
        # (gloo rehearsal with graphs, or after a refused capture) everything stays on the one chain
        side_ok = self._overlap_ok and (not self.use_graph or (getattr(dp, "capturable", False) and self._one_graph_ok is not False))

        def st():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def sort_bucket(ids, begin):
            ws = opt.begin_step_sort(ids, emb.offsets, B, F, D, "local", err=emb.err, begin=begin)   # [++step,] index, sort
            L.launch("cdc_shard_bucket", lib.cdc_shard_bucket,
                     (ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), self.send_ids.data_ptr(), self.slot_of.data_ptr(),
                      self.overflow.data_ptr(), B, F, N, cap), st())

        def stage0():
            sort_bucket(emb.ids, begin=not ahead)

        def stage0_next():
            sort_bucket(self.ids_next, begin=False)

        def serve():
            opt.table_catchup_rows(recv_cur, Bv, F, D, "owner", runs=N, flush=False)   # each sender's list is sorted
            if side_ok:                               # the slice of the owned rows in the background, until the row update
                main, side = torch.cuda.current_stream(), self._side_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    opt.flush_slice(background_waves=self._overlap_waves)
            L.launch("cdc_embed_gather_fwd(owner)", lib.cdc_embed_gather_fwd,
                     (recv_cur.data_ptr(), self.zero_offsets.data_ptr(), opt.table.data_ptr(), self.rows_send.data_ptr(),
                      None, None, Bv, F, D, opt.table.shape[0]), st())

        emb_shadow = plan.shadow_convert_step(emb.out) if (plan.use_g2 and plan.has_shadow(emb.out)) else None

        def expand():
            ws = opt._workspace(B, F, D, "local")
            L.launch("cdc_shard_expand", lib.cdc_shard_expand,
                     (self.rows_recv.data_ptr(), ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), ws["seg"].data_ptr(),
                      ws["perm"].data_ptr(), self.slot_of.data_ptr(), emb.out.ptr, B, F, D, N, cap), st())
            if emb_shadow is not None:                    # the gather launch this replaces also wrote the embeddings' bf16 shadow
                emb_shadow(st())

        def pack():
            ws = opt._workspace(B, F, D, "local")
            opt._segment_sum(ws, emb.out.grad.root, B, F, D, st())
            L.launch("cdc_shard_pack", lib.cdc_shard_pack,
                     (ws["rowgrad"].data_ptr(), ws["uniq"].data_ptr(), ws["cnt"].data_ptr(), self.slot_of.data_ptr(),
                      self.grads_send.data_ptr(), B, F, D, N, cap), st())

        pending = {}

        def exchange_rows_start():
            # the row gradients travel while the grad-weight launches (which nothing here depends on) run
            pending["grads"] = dp.all_to_all_start(self.grads_recv, self.grads_send)

        def exchange_finish():
            dp.wait(pending.pop("grads", None))
            dp.all_reduce_sum(self.arena_and_loss)                      # dense gradients + the loss scalar behind them
            if prefetch:                                                # the next step's row lists travel under this step's updates
                self._ids_pending = dp.all_to_all_start(recv_nxt, self.send_ids)

        def ids_await():
            dp.wait(self.__dict__.pop("_ids_pending", None))

        def update():
            if side_ok:
                torch.cuda.current_stream().wait_stream(self._side_stream())      # the slice (and the look-ahead sort) are done
            if side_ok:
                # the owner's row update and the dense parameters in one launch, as on one GPU (optim.rows_and_dense_step)
                opt.rows_and_dense_step(recv_cur, self.grads_recv, Bv, F, D, plan.param_grads, plan._param_refs, plan.grad_slabs, tag="owner",
                                        short_segments=True)
                self._reg()
                return
            opt.table_step(recv_cur, self.grads_recv, Bv, F, D, "owner", short_segments=True)
            opt.flush_slice()                           # off the rows-exchange critical path: after the owner's update
            opt.dense_step(plan.param_grads, plan._param_refs, plan.grad_slabs)
            self._reg()

        def stage0_next_side():
            # the next batch's sort + bucketing beside the grad-weight launches (its workspaces are free once `pack` has run)
            main, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                sort_bucket(self.ids_next, begin=False)

        def join_side():
            torch.cuda.current_stream().wait_stream(self._side_stream())

        def run_steps(steps):
            def fn():
                s_ = st()
                for step in steps:
                    step(s_)
            return fn

        if have:
            seq = [(True, ids_await), (False, serve)]
        else:
            seq = [(False, stage0), (True, lambda: dp.all_to_all(recv_cur, self.send_ids)), (False, serve)]
        seq += [(True, lambda: dp.all_to_all(self.rows_recv, self.rows_send)), (False, expand)]
        fwd = [s_ for s_ in plan.fwd_steps if s_ is not emb.fwd_step]
        for is_comm, steps in plan.segments(fwd):
            seq.append((is_comm, run_steps(steps)))
        seq.append((False, self._bce))
        late = set(id(s_) for s_ in plan.deferred_dw_steps)
        for is_comm, steps in plan.segments([s_ for s_ in plan.bwd_steps if id(s_) not in late]):
            seq.append((is_comm, run_steps(steps)))
        if prefetch and side_ok:
            seq += [(False, pack), (False, stage0_next_side), (True, exchange_rows_start), (False, run_steps(plan.deferred_dw_steps)),
                    (False, join_side)]
        else:
            seq += [(False, pack), (True, exchange_rows_start), (False, run_steps(plan.deferred_dw_steps))]
            if prefetch:
                seq.append((False, stage0_next))
        seq += [(True, exchange_finish), (False, update)]
        merged = []
        for is_comm, fn in seq:
            if merged and not is_comm and not merged[-1][0]:
                merged[-1][1].append(fn)
            else:
                merged.append((is_comm, [fn]))
        return merged

    def gather_table(self):
        """Row-sharded table: bring every owner's rows up to date and give every rank the whole table (and its Adam
        moments) — before state_dict(), evaluation, or a switch to another trainer.  Host-synchronising collectives."""
        opt = self.opt
        opt.flush_table()
        if not self.dp_on or self.table_dist != "sharded":
            return
        R = opt.table.shape[0]
        own = (torch.arange(R, device=self.device) % self.world == self.dist.rank).to(torch.float32).unsqueeze(1)
        for t in (opt.table.data, opt.table_m, opt.table_v):
            t.mul_(own)                                   # x + 0 + ... + 0 is exact: the all-reduce is a row broadcast
            self.dist.all_reduce_sum(t)
        opt.table_last.fill_(int(opt.step_dev.item()))

    def _ahead_dp(self):
        """Data parallel, row-sharded table: the local sort, the bucketing and the id exchange of batch t+1 leave step t+1's
        critical path when the caller names the next batch (step(..., next_X=)); see _dp_sequence_sharded.  sort_ahead=False: off;
        sort_ahead="force": also with a forced one-rank group."""
        if getattr(self, "_ahead_dp_ok", None) is None:
            # (a forced one-rank group has no transfer for the moved work to hide under — measured 0.803 -> 0.818 ms/step on the
            #  one-GPU box, the step there is bound by the host-side issue of its segments — so it takes part only with
            #  sort_ahead="force", which the RCCL call-path test sets)
            want = self._sort_ahead_wanted
            self._ahead_dp_ok = (self.dp_on and self.table_dist == "sharded" and bool(want) and (self.world > 1 or want == "force"))
            if self._ahead_dp_ok:
                self.ids_next = torch.zeros_like(self.emb.ids)
                self._parity, self._sorted_for, self._dp_seqs = 0, None, {}
        return self._ahead_dp_ok

    def _step_dp(self, X=None, nx=None):
        key = None
        if getattr(self, "_ahead_dp_ok", False):
            xk = (X.data_ptr(), X._version) if torch.is_tensor(X) else None
            have = self._sorted_for is not None and self._sorted_for[:2] == xk
            if not have:
                self.__dict__.pop("_ids_pending", None)                 # (an exchange started for a batch that did not come: its
            prefetch = nx is not None                                   #  buffer is simply overwritten by the next one)
            p = self._parity
            key = (have, prefetch, p)
            if prefetch:
                self._parity = 1 - p
                self._sorted_for = (nx.data_ptr(), nx._version, nx)
            else:
                self._sorted_for = None

        def sequence():
            if key is not None:
                seq = self._dp_seqs.get(key)
                if seq is None:
                    seq = self._dp_seqs[key] = self._dp_sequence_sharded(ahead=True, have=key[0], prefetch=key[1], p=key[2])
                return seq
            if self._dp_seq is None:
                self._dp_seq = self._dp_sequence_sharded() if self.table_dist == "sharded" else self._dp_sequence()
            return self._dp_seq

        seq = sequence()
        if self.use_graph and self._warm >= 2 and self._one_graph_ok is not False and getattr(self.dist, "capturable", False):
            # ONE graph for the whole step, collectives included (round 4: RCCL collectives capture and replay on this stack; what hung
            # in rounds 2-3 was destroy_process_group() with such a graph alive — DataParallel.close() releases them first).  The
            # five-segment cut (launch segments replayed, collectives issued by the host in between) is the fallback
            g = self._step_graphs.get(key)
            if g is None:
                # capture from a quiescent state: an exchange the previous (eager) step left in flight is awaited OUTSIDE the capture
                # (inside, ids_await then finds nothing: a replayed predecessor has joined its own exchange at its end), the device is
                # drained, and the RCCL watchdog thread gets time to retire the eager collectives it is still polling — it must not
                # query events while this thread captures (seen once in a full-suite run: the capture ended "unjoined" and the
                # watchdog then died on an event "last recorded in a capturing stream")
                import time
                self.dist.wait(self.__dict__.pop("_ids_pending", None))
                torch.cuda.synchronize()
                time.sleep(0.25)
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        for _, fns in seq:
                            for fn in fns:
                                fn()
                        # an exchange started for the NEXT step (look-ahead ids) has to be joined inside the graph that started it
                        self.dist.wait(self.__dict__.pop("_ids_pending", None))
                except Exception as e:  # noqa: BLE001
                    import warnings
                    warnings.warn(f"hipGraph capture of the whole data-parallel step failed ({e}); falling back to launch segments")
                    torch.cuda.synchronize()
                    self._one_graph_ok = False
                    self.__dict__.pop("_ids_pending", None)
                    g = None
                    # the sequences were built with a side chain, which launch segments cannot hold: build them again without
                    self._dp_seq = None
                    if hasattr(self, "_dp_seqs"):
                        self._dp_seqs.clear()
                    seq = sequence()
                else:
                    if self._one_graph_ok is None:
                        self._one_graph_ok = True
                        self.dist._on_close.append(self._drop_graphs)
                    self._step_graphs[key] = g
            if g is not None:
                g.replay()
                self._warm += 1
                return
        for i, (is_comm, fns) in enumerate(seq):
            if is_comm or not (self.use_graph and self._warm >= 2):
                for fn in fns:
                    fn()
                continue
            i = (key, i)
            g = self._stage_graphs.get(i)
            if g is None:
                # thread-local capture mode: the RCCL watchdog thread keeps querying events while we capture
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        for fn in fns:
                            fn()
                except Exception as e:  # noqa: BLE001  (capture refused: run this and every later segment eagerly)
                    import warnings
                    warnings.warn(f"hipGraph capture of launch segment {i} failed ({e}); continuing without graphs")
                    torch.cuda.synchronize()
                    self.use_graph = False
                    self._stage_graphs.clear()
                    for fn in fns:
                        fn()
                    continue
                self._stage_graphs[i] = g
            g.replay()
        self._warm += 1

    def step(self, X, y, group=None, next_X=None):
        """One training step. X int32 [B,F]; y int16/float [B] or [B,1]; group int64 [B] or [B,1] (multi mode).
        next_X (optional, single GPU): the ids of the batch the NEXT call will be given (device int32 [B,F], not modified until
        then) — its rows are sorted beside this step's forward/backward (_sort_ahead); a next call with another tensor simply sorts
        its own.  Returns (bce_loss, reg_loss) as device tensors (no host synchronisation)."""
        gdst = self.group if self.group is not None else (self.group_in if self.mode == "single_group" else None)
        yf = y.reshape(-1)
        gf = None if (group is None or gdst is None) else group.reshape(-1)
        fast = (X.is_cuda and X.dtype == torch.int32 and X.is_contiguous() and tuple(X.shape) == (self.B, self.emb.F) and
                yf.is_cuda and yf.dtype == torch.int16 and yf.is_contiguous() and yf.numel() == self.B and
                (gdst is None or (gf is not None and gf.is_cuda and gf.dtype == torch.int64 and gf.is_contiguous() and gf.numel() == self.B)))
        if not self._reg_checked or (self.opt.table_mode == "lazy" and not self.opt.table_reg_ready):
            # the first reported loss of a lazy run carries the table's L2 term like every later one (a freshly built or LOADED
            # optimiser has not evaluated it yet: load_state_dict clears table_reg_ready, also for a TrainStep that already ran).
            # Before the staging launch: with the look-ahead sort that launch advances the step counter, and the refresh brings
            # the table to the counter's step
            self._reg_checked = True
            if self.opt.table_mode == "lazy" and not self.opt.table_reg_ready:
                self.refresh_table_reg()
        ahead = (not self.dp_on) and self._sort_ahead() and self._overlap()      # (profile(overlap=False) switches the side chain off)
        if not ahead and getattr(self, "_ahead_ok", False):
            self._sorted_for = None
        ahead_dp = self.dp_on and self._ahead_dp()
        ahead = ahead or ahead_dp
        nx = None
        if fast and ahead:                                        # + the next batch's ids for the look-ahead sort, + begin_step
            if (next_X is not None and next_X.is_cuda and next_X.dtype == torch.int32 and next_X.is_contiguous() and
                    tuple(next_X.shape) == (self.B, self.emb.F)):
                nx = next_X
            L.launch("cdc_stage_batch", self.lib.cdc_stage_batch_next,
                     (X.data_ptr(), yf.data_ptr(), None if gdst is None else gf.data_ptr(), self.emb.ids.data_ptr(), self.y.data_ptr(),
                      None if gdst is None else gdst.data_ptr(), self.B, self.emb.F, None if nx is None else nx.data_ptr(),
                      self.ids_next.data_ptr(), self.opt.step_dev.data_ptr(), self.opt.reg_sum.data_ptr(), 2,
                      self.field_dims_dev.data_ptr(), self.alias.data_ptr()),
                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
        elif fast:                                                # one launch instead of three copies
            L.launch("cdc_stage_batch", self.lib.cdc_stage_batch,
                     (X.data_ptr(), yf.data_ptr(), None if gdst is None else gf.data_ptr(), self.emb.ids.data_ptr(), self.y.data_ptr(),
                      None if gdst is None else gdst.data_ptr(), self.B, self.emb.F, self.field_dims_dev.data_ptr(), self.alias.data_ptr()),
                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
        else:
            self.emb.ids.copy_(X)
            bad = ((self.emb.ids < 0) | (self.emb.ids >= self.field_dims_dev)).reshape(-1).to(torch.int32)
            torch.maximum(self.alias, (bad * torch.arange(1, bad.numel() + 1, device=bad.device, dtype=torch.int32)).max().reshape(1), out=self.alias)
            self.y.copy_(yf)
            if gdst is not None:
                gdst.copy_(gf)
            if ahead:
                self.opt.begin_step()
        if self.dp_on:
            self._step_dp(X if fast else None, nx)
        elif ahead:
            self._step_ahead(X if fast else None, nx)
        elif self.use_graph and self._warm >= 2:
            if self.graph is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch_all()
                self.graph = g
            self.graph.replay()
        else:
            self._launch_all()
            self._warm += 1
        return self.loss, self.reg

    def profile(self, batches, n_steps=10, skip=2, overlap=True):
        """Per-launch timing of `n_steps` eager steps (HIP events on the stream each launch goes to); the first `skip` are not
        counted.  Returns {name: {"ms_per_step", "launches_per_step", "flops_per_step", "bytes_per_step"}}.  Events are read back
        and released every few steps: the runtime backs each timed event with a signal from a bounded pool.
        overlap=False: the whole step on one stream, i.e. every kernel's duration with the chip to itself."""
        was_graph, self.use_graph = self.use_graph, False
        self._overlap()
        was_overlap, self._overlap_ok = self._overlap_ok, bool(self._overlap_ok and overlap)
        rec = []
        out = {}
        used = max(n_steps - skip, 1)

        def fold(count):
            torch.cuda.synchronize()
            if count:
                for name, e0, e1, fl, nb in rec:
                    d = out.setdefault(name, {"ms_per_step": 0.0, "launches_per_step": 0.0, "flops_per_step": 0.0, "bytes_per_step": 0.0})
                    d["ms_per_step"] += e0.elapsed_time(e1) / used
                    d["launches_per_step"] += 1.0 / used
                    d["flops_per_step"] += fl / used
                    d["bytes_per_step"] += nb / used
            rec.clear()

        L.PROFILE = rec
        try:
            for i in range(n_steps):
                if i == skip:
                    fold(False)
                self.step(*batches[i % len(batches)])
                if i >= skip and (i - skip) % 4 == 3:
                    fold(True)
            fold(n_steps > skip)
        finally:
            L.PROFILE = None
            self.use_graph = was_graph
            self._overlap_ok = was_overlap
        return out

    def sibling(self, batch_size, global_rows=None, cap_rows=None):
        """A TrainStep for another batch size on the SAME model and optimiser state (the ragged last batch of an epoch,
        which the reference trains on like any other: run.py:476); runs eagerly.  Data parallel: batch_size is THIS rank's share
        of the ragged global batch, global_rows its true size, cap_rows the largest share (ceil(global_rows / world)); every rank
        must call it (the step's collectives).  Row-sharded table, or equal shares with the replicated one."""
        key = (int(batch_size), None if global_rows is None else int(global_rows), None if cap_rows is None else int(cap_rows))
        sib = self.__dict__.setdefault("_siblings", {})
        ts = sib.get(key)
        if ts is None:
            if self.world > 1:
                if global_rows is None or cap_rows is None:
                    raise ValueError("a ragged batch under data parallelism needs global_rows and cap_rows (data.train_epoch passes them)")
                if self.table_dist != "sharded" and int(global_rows) != int(batch_size) * self.world:
                    raise NotImplementedError("uneven shares of a ragged batch need the row-sharded table (the replicated table's "
                                              "all-gathers are equal-sized)")
                ts = TrainStep(self.model, self.opt, int(batch_size), mode=self.mode, use_graph=False, dist=self.dist, sync_bn=self.sync_bn,
                               table_dist=self.table_dist, shard_slack=self._shard_slack, train_mode=self.train_mode, sort_ahead=False,
                               global_rows=int(global_rows), cap_rows=int(cap_rows))
            else:
                ts = TrainStep(self.model, self.opt, int(batch_size), mode=self.mode, use_graph=False, train_mode=self.train_mode)
            sib[key] = ts
        return ts

    def check_ids(self):
        """Host-synchronising: raises IndexError like the reference if the last batch held an out-of-range id."""
        bad = int(self.emb.err.item())
        if bad:
            self.emb.err.zero_()
            self.alias.zero_()
            raise IndexError(f"index out of range in self (flat position {bad - 1})")
        al = int(self.alias.item())
        if al:
            self.alias.zero_()
            b, f = divmod(al - 1, self.emb.F)
            raise ValueError(f"the id at batch position {b}, field {f} lies outside the field's vocabulary but inside the table: it aliases a row "
                             "of another field.  The reference trains that row through both fields at once (one Adam update of the summed "
                             "gradient, model/layer.py:152-153); the per-field row lists of this path update it once per field, so the "
                             "results of that step are not the reference's")
        for op in self.plan.ops:                                  # in-launch exchanges that gave up (csrc/tower.hip)
            word = getattr(op, "tmo_word", None)
            if torch.is_tensor(word) and int(word.item()) & L.TOWER_ERR_TIMEOUT:
                word.zero_()
                getattr(op, "ws").zero_()
                raise RuntimeError("the fused tower launch timed out waiting for its other workgroups (BatchNorm statistics "
                                   "exchange); the results of that step are void")
        if self.dp_on and self.table_dist == "sharded":
            over = int(self.overflow.item())
            if over:
                self.overflow.zero_()
                raise RuntimeError(f"row-sharded table: a per-owner row list needed {over} slots but holds {self.cap}; "
                                   f"raise shard_slack (rows beyond the capacity were dropped from that step)")
