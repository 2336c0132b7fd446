"""Static launch plans: the forward and backward kernel sequences of one model at one batch size.

A plan owns every activation / gradient buffer (allocated once through torch's caching allocator) and
every pre-filled C-ABI argument block, so running it is nothing but a fixed list of asynchronous
launches on the current HIP stream — the whole step is hipGraph-capturable and costs no Python work
per element.  The model classes in model/ describe their dataflow with the small builder below; the
reference does the same work through eager ATen ops and autograd (model/layer.py, ple.py, mmoe.py, ...).

Gradient conventions
  * `Buf.grad` of a buffer produced by a linear layer with a fused activation (relu / dropout) holds
    the gradient w.r.t. the PRE-activation value: consumers apply the producer's mask while writing
    (glinear_bwd_x / gate_pool_bwd epilogues), so no separate mask pass exists.
  * a BatchNorm output's grad is w.r.t. its post-activation value (bn_bwd applies its own mask).
  * several consumers of one buffer accumulate: the first writer (in backward order) stores, later
    ones add; the builder works the flags out when the plan is finalised.
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Buf:
    """fp32 2-D view [rows, cols] with row stride `ld` (elements) inside a root tensor."""

    def __init__(self, root, rows, cols, ld=None, col0=0, plan=None):
        self.root = root                  # torch tensor [rows, ld_root]
        self.rows = rows
        self.cols = cols
        self.ld = ld if ld is not None else root.stride(0)
        self.col0 = col0
        self.plan = plan
        self.mask = None                  # (scale, n_cols): relu/dropout fused into the producer covers cols [0,n_cols)
        self.is_param_like = False

    @property
    def ptr(self):
        return self.root.data_ptr() + 4 * self.col0

    def cptr(self):
        return C.c_void_p(self.ptr)

    def slice(self, c0, c1):
        b = Buf(self.root, self.rows, c1 - c0, self.ld, self.col0 + c0, self.plan)
        if self.mask is not None:
            scale, act_cols = self.mask
            if c0 < act_cols:
                b.mask = (scale, min(act_cols, c1) - c0)
        return b

    def tensor(self):
        return self.root[:, self.col0:self.col0 + self.cols]

    # ---- gradient view (same geometry inside the root's gradient tensor) ----
    @property
    def grad(self):
        g = self.plan._root_grad(self.root)
        return Buf(g, self.rows, self.cols, g.stride(0), self.col0, self.plan)


class PView:
    """One index of the leading dimension of a Parameter (e.g. expert k of a [n_expert, E, r] tensor), usable
    wherever an op takes a weight: same .shape / .data_ptr() / .numel() surface; its gradient is the matching
    slice of the parameter's gradient."""

    def __init__(self, param, index):
        self.param, self.index = param, index
        self.shape = param.shape[1:]

    def data_ptr(self):
        return self.param.data[self.index].data_ptr()

    def numel(self):
        return self.param.data[self.index].numel()

    @property
    def data(self):
        return self.param.data[self.index]


class TView:
    """A plain tensor used as a weight, with its own gradient tensor (STAR's fused W_d*W_s etc.): not a parameter,
    so it never shows up in plan.param_grads."""

    def __init__(self, tensor, grad):
        self.tensor, self.grad_tensor = tensor, grad
        self.shape = tensor.shape

    def data_ptr(self):
        return self.tensor.data_ptr()

    def numel(self):
        return self.tensor.numel()

    @property
    def data(self):
        return self.tensor


class _GradState:
    """Tracks which column intervals of a root gradient tensor the backward sequence has initialised."""

    def __init__(self, plan=None):
        self.done = {}      # id(root) -> list of (c0, c1)
        self.plan = plan

    def claim(self, buf):
        """Returns True if `buf`'s gradient region is already initialised (=> accumulate), else marks it."""
        key = id(buf.root)
        iv = self.done.setdefault(key, [])
        c0, c1 = buf.col0, buf.col0 + buf.cols
        if self.plan is not None:
            # whoever claims the region is about to write it: a bf16 shadow of the gradient written earlier is stale from here on
            self.plan.invalidate_shadow(buf.grad)
        for a, b in iv:
            if a <= c0 and c1 <= b:
                return True
        for a, b in iv:
            if not (c1 <= a or b <= c0):
                raise RuntimeError(f"partial gradient overlap on a buffer: [{c0},{c1}) vs [{a},{b})")
        iv.append((c0, c1))
        # merge adjacent column intervals (slices claimed one by one later read as one whole buffer)
        iv.sort()
        merged = [iv[0]]
        for a, b in iv[1:]:
            if a == merged[-1][1]:
                merged[-1] = (merged[-1][0], b)
            else:
                merged.append((a, b))
        iv[:] = merged
        return False

    def is_set(self, buf):
        c0, c1 = buf.col0, buf.col0 + buf.cols
        return any(a <= c0 and c1 <= b for a, b in self.done.get(id(buf.root), []))


class Plan:
    def __init__(self, device, B, precision="bf16", training=True, dropout=0.0, seed=0, step_dev=None, grad_arena=None, dist=None, g2=True,
                 defer_dw_reduce=False):
        assert precision in ("bf16", "f32")
        self.lib = L.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.HipExtensionError("the HIP hot path needs a GPU device (got %s); there is no CPU fallback" % device)
        self.B = B
        self.prec = L.PREC_BF16 if precision == "bf16" else L.PREC_F32
        self.training = training
        self.dropout = float(dropout) if training else 0.0
        self.keep_scale = 1.0 / (1.0 - self.dropout) if self.dropout > 0 else 1.0
        self.seed = seed
        self.ops = []
        self.fwd_steps = []
        self.bwd_steps = []
        self._grads = {}                  # id(root) -> grad tensor
        self._roots = {}
        self.param_grads = {}             # id(param) -> grad tensor
        self._param_refs = {}
        self._param_grad_set = set()
        self._op_count = 0
        # device step counter: offsets the dropout stream per step (shared with the optimiser when given)
        self.step_dev = step_dev if step_dev is not None else torch.zeros(1, dtype=torch.int32, device=self.device)
        # optional flat fp32 arena the dense-parameter gradients are carved from (one all-reduce under DP)
        self.grad_arena = grad_arena
        self._arena_used = 0
        # data parallel: BatchNorm statistics are exchanged between ranks (global-batch statistics, like the reference's
        # single process); the exchange steps are marked so the step driver can cut its launch graphs around them
        self.dist = dist if (dist is not None and (dist.world_size > 1 or getattr(dist, "force", False))) else None
        self._bn_ws = None
        self._bn_ws_need = 0
        self._rowdot_ws = None
        self._rowdot_ws_need = 0
        self._gemm_ws = None
        self._gemm_ws_need = 0
        self._gemm_ws_users = []          # argument blocks whose .workspace is the grad-weight split-K buffer
        self._lin_producer = {}           # linear output (root, col0, cols) -> its launch group (BatchNorm statistics fusion)
        self._last_bn_step = 0
        self._wt = {}                     # weight key -> (weight object, transposed copy [K,N]) refreshed at the start of backward
        self._deferred_dw = []            # grad-weight groups of every layer, launched together at the end of backward
        # defer_dw_reduce (the single-GPU training step): the batched grad-weight launches leave their split-K slabs unreduced and
        # the dense Adam launch adds them while it reads the gradient (cdc_adam_tensor.slabs): two launches and a round trip of
        # the gradients through memory less per step.  grad_slabs: address of a parameter gradient -> (first slab, stride, count);
        # the gradient tensors of those parameters are NOT written by backward()
        self.defer_dw_reduce = bool(defer_dw_reduce)
        self.grad_slabs = {}
        self.finalized = False
        self.loss_inputs = None
        # bf16 contraction path with operands that are bf16 in memory (csrc/gemm2.hip): shadows of activations, gradients, weights
        # g2=False (tests only): every bf16 contraction through csrc/gemm.hip's register-staged path — the one ragged launches
        # (STAR's per-domain groups) take — so that the two paths can be held against each other
        self.use_g2 = self.prec == L.PREC_BF16 and bool(g2)
        self._shadows = {}                # id(fp32 root) -> (root, bf16 tensor [rows64, cols64 + 64], zero padded)
        self._sh_have = {}                # id(root) -> [(c0, c1)] columns whose shadow is current at this point of the sequence
        self._sh_wanted = {}              # id(root) -> [(c0, c1)] columns some contraction reads through the shadow
        self._wsh = {}                    # weight key -> (weight, straight copy [N, K64], transposed copy [K, N64])
        self._half_only = set()           # id(root): activations (and their gradients) that exist ONLY as bf16 shadows

    # ---------------------------------------------------------------- buffers
    def new(self, cols, rows=None, dtype=torch.float32):
        rows = self.B if rows is None else rows
        t = torch.empty((rows, cols), dtype=dtype, device=self.device)
        self._roots[id(t)] = t
        return Buf(t, rows, cols, plan=self)

    def _root_grad(self, root):
        g = self._grads.get(id(root))
        if g is None:
            g = torch.zeros_like(root)
            self._grads[id(root)] = g
            self._roots[id(root)] = root
        return g

    def param_grad(self, p):
        if isinstance(p, TView):
            return p.grad_tensor
        if isinstance(p, PView):
            return self.param_grad(p.param)[p.index]
        g = self.param_grads.get(id(p))
        if g is None:
            n = p.numel()
            if self.grad_arena is not None and self._arena_used + n <= self.grad_arena.numel():
                g = self.grad_arena[self._arena_used:self._arena_used + n].view(p.shape)
                g.zero_()
                self._arena_used += (n + 3) // 4 * 4          # keep every slice 16-byte aligned
            else:
                g = torch.zeros_like(p.data)
            self.param_grads[id(p)] = g
            self._param_refs[id(p)] = p
        return g

    def ensure_grad(self, buf, gs):
        """Make sure buf.grad is initialised in the backward sequence (zero it if nobody wrote it)."""
        if not gs.is_set(buf):
            t = buf.grad.tensor()
            self.bwd_steps.append(lambda s, t=t: t.zero_())
            gs.claim(buf)

    def _claim_param(self, p):
        key = (id(p.param), p.index) if isinstance(p, PView) else (id(p.tensor) if isinstance(p, TView) else id(p))
        if key in self._param_grad_set:
            return True
        self._param_grad_set.add(key)
        return False

    def _next_seed(self):
        self._op_count += 1
        return (self.seed * 1000003 + self._op_count * 7919) & 0xFFFFFFFFFFFFFFFF

    def need_bn_ws(self, rows, total_c):
        self._bn_ws_need = max(self._bn_ws_need, 2 * math.ceil(rows / L.BN_ROWS_PER_BLOCK) * total_c)

    def wt_of(self, w, half=False):
        """per-step transposed copy [K,N] of a linear weight [N,K]: grad-input then streams both operands with the
        reduction index contiguous (no transposing LDS stores).  half: the copy is written as bf16 (what the bf16 MFMA
        path would round the fp32 copy to while staging it) — half the operand bytes, no conversion in the loop."""
        key = (id(w.param), w.index) if isinstance(w, PView) else (id(w.tensor) if isinstance(w, TView) else id(w))
        key = (key, bool(half))
        hit = self._wt.get(key)
        if hit is None:
            N, K = w.shape
            hit = (w, torch.empty((K, N), dtype=torch.bfloat16 if half else torch.float32, device=self.device), bool(half))
            self._wt[key] = hit
        return hit[1]

    def _emit_transposes(self):
        steps = []
        items = list(self._wt.values())
        for c0 in range(0, len(items), L.MAX_TENSORS):
            chunk = items[c0:c0 + L.MAX_TENSORS]
            a = L.TransposeArgs()
            a.n = len(chunk)
            mask = 0
            for i, (w, wt, half) in enumerate(chunk):
                a.t[i].src, a.t[i].dst = w.data_ptr(), wt.data_ptr()
                a.t[i].rows, a.t[i].cols = w.shape[0], w.shape[1]
                if half:
                    mask |= 1 << i
            a.bf16_mask = mask
            self._keep_args = getattr(self, "_keep_args", []) + [a]
            steps.append(self.call("cdc_transpose_multi", C.byref(a)))
        return steps

    def _emit_deferred_dw(self):
        """All layers' grad-weight contractions are independent of the rest of backward once their dZ exists: they go out
        together at the end, in launches that fill the chip (instead of one under-filled launch per layer)."""
        groups = self._deferred_dw
        # narrow outputs (N <= 64: gates, the last expert level, towers) go to launches of their own: the grad-weight kernel then
        # uses 64x64 tiles for them instead of padding each to 128 rows (a gate with 4 outputs filled 3 % of its tiles).  The
        # order inside a class is kept (a group that accumulates follows its base; both have the same N)
        groups = ([g for g in groups if g.get("dzh") is not None and g["N"] > 64] +
                  [g for g in groups if g.get("dzh") is not None and g["N"] <= 64] +
                  [g for g in groups if g.get("dzh") is None])
        launches, cur = [], []
        narrow = lambda g: g.get("dzh") is not None and g["N"] <= 64      # noqa: E731
        for g in groups:
            # (a launch is either all-shadows (csrc/gemm2.hip) or all-fp32-operands: layers too narrow for the bf16 tiles sit
            # next to wide ones in small models)
            if cur and (len(cur) >= L.MAX_GROUPS or g["accumulate"] or (g.get("dzh") is None) != (cur[0].get("dzh") is None) or
                        narrow(g) != narrow(cur[0])):
                launches.append(cur)
                cur = []
            cur.append(g)
        if cur:
            launches.append(cur)
        steps = []
        for chunk in launches:
            a = L.LinBwdwArgs()
            a.n_groups = len(chunk)
            Mmax = max(g["M"] for g in chunk)
            if any(g.get("needs_shadows") for g in chunk) and not all(g.get("dzh") is not None for g in chunk):
                raise RuntimeError("a grad-weight launch mixes shadow-only activations with groups that have no shadows")
            if all(g.get("dzh") is not None for g in chunk):
                # shadows (csrc/gemm2.hip k_g2_tn): 128x128 output tiles unless every group is at most 64x64; a row slice of at least
                # four 64-row slabs per workgroup
                T = 64 if all(g["N"] <= 64 for g in chunk) else 128
                tiles = sum(math.ceil(g["N"] / T) * math.ceil(g["K"] / T) for g in chunk)
                S = max(1, min(512 // max(tiles, 1), max(Mmax // 256, 1), 32))
                if self.defer_dw_reduce:                   # the Adam launch reads the slabs in rounds of eight (csrc/rowops.hip)
                    S = min(S, 8)
            else:
                tiles = sum(math.ceil(g["N"] / 64) * math.ceil(g["K"] / 64) for g in chunk)
                S = max(1, min(1024 // max(tiles, 1), max(Mmax // 128, 1), 64))       # 1024: measured best of 384…2560 at C2
            a.split_k = S
            if S > 1:
                self.need_gemm_ws(S * sum(g["N"] * g["K"] + g["N"] for g in chunk))
            a.row_offsets = None
            for i, g in enumerate(chunk):
                G = a.g[i]
                G.dz, G.lddz, G.x, G.ldx = g["dz"], g["lddz"], g["x"], g["ldx"]
                G.dw, G.lddw, G.db = g["dw"], g["lddw"], g["db"]
                G.M, G.N, G.K, G.accumulate = g["M"], g["N"], g["K"], g["accumulate"]
                if g.get("dzh") is not None:
                    (G.dzh, G.lddzh), (G.xh, G.ldxh) = g["dzh"], g["xh"]
                else:
                    G.dzh = G.xh = None
            self._keep_args = getattr(self, "_keep_args", []) + [a]
            fl = sum(2.0 * g["M"] * g["N"] * g["K"] for g in chunk)
            steps.append((a, fl))
        return steps

    def need_gemm_ws(self, n):
        self._gemm_ws_need = max(self._gemm_ws_need, n)

    def need_rowdot_ws(self, n):
        self._rowdot_ws_need = max(self._rowdot_ws_need, n)

    # ---------------------------------------------------------------- bf16 shadows (gemm2)
    def _shadow_root(self, root):
        hit = self._shadows.get(id(root))
        if hit is None:
            rows, cols = root.shape
            t = torch.zeros(((rows + 63) // 64 * 64, (cols + 63) // 64 * 64 + 128), dtype=torch.bfloat16, device=self.device)
            hit = (root, t)
            self._shadows[id(root)] = hit
        return hit[1]

    def shadow_view(self, buf):
        """(address, row stride in elements) of buf's columns inside the bf16 shadow of its root"""
        t = self._shadow_root(buf.root)
        return t.data_ptr() + 2 * buf.col0, t.stride(0)

    @staticmethod
    def _iv_covered(table, buf):
        c0, c1 = buf.col0, buf.col0 + buf.cols
        return any(a <= c0 and c1 <= b for a, b in table.get(id(buf.root), []))

    @staticmethod
    def _iv_add(table, buf):
        iv = table.setdefault(id(buf.root), [])
        iv.append((buf.col0, buf.col0 + buf.cols))
        iv.sort()
        merged = [iv[0]]
        for a, b in iv[1:]:
            if a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        iv[:] = merged

    def want_shadow(self, buf):
        self._iv_add(self._sh_wanted, buf)

    def shadow_wanted(self, buf):
        c0, c1 = buf.col0, buf.col0 + buf.cols
        return any(a < c1 and c0 < b for a, b in self._sh_wanted.get(id(buf.root), []))

    def mark_shadow(self, buf):
        self._iv_add(self._sh_have, buf)

    def has_shadow(self, buf):
        return self._iv_covered(self._sh_have, buf)

    def invalidate_shadow(self, buf):
        iv = self._sh_have.get(id(buf.root))
        if iv:
            c0, c1 = buf.col0, buf.col0 + buf.cols
            iv[:] = [(a, b) for a, b in iv if not (a < c1 and c0 < b)]

    def ensure_shadows(self, bufs, steps):
        """one conversion launch (appended to `steps`) for those of `bufs` whose shadow no producer has written"""
        todo = []
        for b in bufs:
            if not self.has_shadow(b) and not any(t.root is b.root and t.col0 == b.col0 and t.cols == b.cols for t in todo):
                todo.append(b)
        for c0 in range(0, len(todo), L.MAX_GROUPS):
            chunk = todo[c0:c0 + L.MAX_GROUPS]
            a = L.ShadowArgs()
            a.n = len(chunk)
            for i, b in enumerate(chunk):
                T = a.t[i]
                T.src, T.ld_src = b.ptr, b.ld
                T.dst, T.ld_dst = self.shadow_view(b)
                T.rows, T.cols = b.rows, b.cols
                self.mark_shadow(b)
            self._keep_args = getattr(self, "_keep_args", []) + [a]
            steps.append(self.call("cdc_shadow_bf16", C.byref(a)))

    def make_half_only(self, buf):
        """The buffer's values are only ever read by bf16 contractions (hidden activations of a BatchNorm-free MLP stack): keep
        them — and their gradient — as bf16 shadows alone.  The fp32 tensors stay allocated (addresses are part of many
        argument blocks) but are never written; they are poisoned so that a consumer this rule did not foresee fails loudly."""
        self._half_only.add(id(buf.root))
        self._half_only.add(id(self._root_grad(buf.root)))
        buf.root.fill_(float("nan"))
        self._root_grad(buf.root).fill_(float("nan"))

    def is_half_only(self, buf):
        return id(buf.root) in self._half_only

    def make_value_half_only(self, buf):
        """like make_half_only for the values alone (their gradient keeps its fp32 form)"""
        self._half_only.add(id(buf.root))
        buf.root.fill_(float("nan"))

    def make_grad_half_only(self, buf):
        """the GRADIENT of buf is only ever read by bf16 contractions (dZ of a linear layer whose grad-input / grad-weight launches
        read the shadow): it is written as bf16 alone"""
        g = self._root_grad(buf.root)
        self._half_only.add(id(g))
        g.fill_(float("nan"))

    def shadow_convert_step(self, buf):
        """a launch step that rewrites buf's shadow from its fp32 values (for a caller that fills buf behind the plan's back:
        the row-sharded trainer replaces the gather launch by its row exchange)"""
        a = L.ShadowArgs()
        a.n = 1
        T = a.t[0]
        T.src, T.ld_src = buf.ptr, buf.ld
        T.dst, T.ld_dst = self.shadow_view(buf)
        T.rows, T.cols = buf.rows, buf.cols
        self._keep_args = getattr(self, "_keep_args", []) + [a]
        return self.call("cdc_shadow_bf16", C.byref(a))

    def wshadow(self, w):
        """bf16 copies of a linear weight [N,K], refreshed once per step ahead of the forward: the straight copy [N, K64]
        (forward's B operand) and the transposed copy [K, N64] (grad-input's B operand), zero padded to whole K-slabs"""
        key = (id(w.param), w.index) if isinstance(w, PView) else id(w)
        hit = self._wsh.get(key)
        if hit is None:
            N, K = w.shape
            h = torch.zeros((N, (K + 63) // 64 * 64), dtype=torch.bfloat16, device=self.device)
            t = torch.zeros((K, (N + 63) // 64 * 64), dtype=torch.bfloat16, device=self.device)
            hit = (w, h, t)
            self._wsh[key] = hit
        return hit[1], hit[2]

    def _emit_wshadows(self):
        steps = []
        items = list(self._wsh.values())
        for c0 in range(0, len(items), L.MAX_TENSORS):
            chunk = items[c0:c0 + L.MAX_TENSORS]
            a = L.WShadowArgs()
            a.n = len(chunk)
            for i, (w, h, t) in enumerate(chunk):
                T = a.t[i]
                T.src = w.data_ptr()
                T.dst_h, T.ld_h = h.data_ptr(), h.stride(0)
                T.dst_t, T.ld_t = t.data_ptr(), t.stride(0)
                T.rows, T.cols = w.shape[0], w.shape[1]
            self._keep_args = getattr(self, "_keep_args", []) + [a]
            steps.append(self.call("cdc_weight_shadows", C.byref(a)))
        return steps

    # ---------------------------------------------------------------- ops
    def add(self, op):
        self.ops.append(op)
        return op

    def finalize(self, outputs):
        """outputs: list of Bufs whose .grad the caller seeds before backward()."""
        self._bn_ws = torch.empty(max(self._bn_ws_need, 1), dtype=torch.float64, device=self.device)
        self._rowdot_ws = torch.empty(max(self._rowdot_ws_need, 1), dtype=torch.float32, device=self.device)
        self._gemm_ws = torch.empty(max(self._gemm_ws_need, 1), dtype=torch.float32, device=self.device)
        gs = _GradState(self)
        for o in outputs:
            gs.claim(o)
            _ = o.grad
        self.outputs = outputs
        for op in self.ops:
            op.build_fwd(self)
        for op in reversed(self.ops):
            op.build_bwd(self, gs)
        dw = self._emit_deferred_dw()
        if self._gemm_ws.numel() < self._gemm_ws_need:
            self._gemm_ws = torch.empty(self._gemm_ws_need, dtype=torch.float32, device=self.device)
        for a in self._gemm_ws_users:            # grad-weight launches built before the batched ones sized the workspace: a pointer
            a.workspace = self._gemm_ws.data_ptr()   # taken earlier would dangle once the buffer has been re-allocated
        self.deferred_dw_steps = []          # the tail of bwd_steps nothing else in the backward depends on
        # launches that leave the slab reduction to the optimiser get a workspace region of their own behind the shared one
        # (their slabs must survive until the dense Adam launch)
        own, extra = {}, 0
        shad = lambda a: all(a.g[i].dzh and a.g[i].xh for i in range(a.n_groups))      # noqa: E731
        splittable = lambda a: a.split_k > 1 and not any(a.g[i].accumulate for i in range(a.n_groups))     # noqa: E731
        # the wide and the narrow class of the batched grad-weight contractions as ONE launch (csrc/gemm2.hip k_g2_tn_dual)?
        two_classes = (len(dw) == 2 and shad(dw[0][0]) and shad(dw[1][0]) and any(dw[0][0].g[i].N > 64 for i in range(dw[0][0].n_groups)) and
                       all(dw[1][0].g[i].N <= 64 for i in range(dw[1][0].n_groups)))
        # ... whose slabs the consumer adds up (single GPU: the dense Adam launch) or, where the reduced gradient is needed at once (data
        # parallelism all-reduces it), ONE more launch (cdc_glinear_bwd_w_pair_reduce): two launches instead of four
        pair_reduce = (not self.defer_dw_reduce) and two_classes and all(splittable(a) for a, _ in dw)
        if self.defer_dw_reduce or pair_reduce:
            for a, fl in dw:
                if splittable(a):
                    slab = sum(a.g[i].N * a.g[i].K + a.g[i].N for i in range(a.n_groups))
                    own[id(a)] = (extra, slab)
                    extra += (a.split_k * slab + 3) // 4 * 4
        if extra:
            shared = (self._gemm_ws_need + 3) // 4 * 4
            self._gemm_ws = torch.empty(shared + extra, dtype=torch.float32, device=self.device)
            for a in self._gemm_ws_users:
                a.workspace = self._gemm_ws.data_ptr()
        pair = None
        if two_classes and all(id(a) in own or a.split_k <= 1 for a, _ in dw) and (self.defer_dw_reduce or pair_reduce):
            pair = (dw[0][0], dw[1][0])
        for a, fl in dw:
            a.workspace = self._gemm_ws.data_ptr()
            if id(a) in own:
                off, slab = own[id(a)]
                a.workspace = self._gemm_ws.data_ptr() + 4 * (shared + off)
                a.defer_reduce = 1
                pos = 0
                for i in range(a.n_groups):
                    G = a.g[i]
                    if self.defer_dw_reduce:
                        self.grad_slabs[G.dw] = (a.workspace + 4 * pos, slab, a.split_k)
                        if G.db:
                            self.grad_slabs[G.db] = (a.workspace + 4 * (pos + G.N * G.K), slab, a.split_k)
                    pos += G.N * G.K + G.N
            if pair is not None:
                continue
            step = self.call("cdc_glinear_bwd_w", C.byref(a), self.prec, flops=fl)
            self.bwd_steps.append(step)
            self.deferred_dw_steps.append(step)
        if pair is not None:
            # the wide and the narrow class of the batched grad-weight contractions in one launch (csrc/gemm2.hip k_g2_tn_dual): the
            # two argument blocks travel as a device copy, made now that their workspaces are final
            aw, an = pair
            self._dw_pair_dev = torch.frombuffer(bytearray(bytes(aw) + bytes(an)), dtype=torch.uint8).to(self.device)
            step = self.call("cdc_glinear_bwd_w_pair_reduce" if pair_reduce else "cdc_glinear_bwd_w_pair", C.byref(aw), C.byref(an),
                             self._dw_pair_dev.data_ptr(), what="cdc_glinear_bwd_w", flops=sum(fl for _, fl in dw))
            self.bwd_steps.append(step)
            self.deferred_dw_steps.append(step)
        self.bwd_steps[0:0] = self._emit_transposes()
        wsh = self._emit_wshadows()
        self.fwd_steps[0:0] = wsh
        self.n_wshadow_steps = len(wsh)      # the head of fwd_steps that reads nothing but the weights (trainer: runs beside the catch-up)
        self.finalized = True

    def comm(self, fn):
        """a collective between launches (runs eagerly; never part of a captured graph)"""
        def run(stream):
            fn()
        run.is_comm = True
        return run

    @staticmethod
    def segments(steps):
        """[(is_comm, [steps...])]: maximal runs of launches, cut at every collective"""
        out = []
        for st in steps:
            c = bool(getattr(st, "is_comm", False))
            if out and out[-1][0] == c and not c:
                out[-1][1].append(st)
            else:
                out.append((c, [st]))
        return out

    def forward(self):
        s = _stream()
        for fn in self.fwd_steps:
            fn(s)

    def backward(self):
        s = _stream()
        for fn in self.bwd_steps:
            fn(s)

    def call(self, fn_name, *args, what=None, flops=0.0, nbytes=0.0):
        fn = getattr(self.lib, fn_name)
        what = what or fn_name
        if os.environ.get("CDC_PROFILE_DETAIL") == "1":          # one timing bucket per launch of the sequence, not per entry point
            self._call_seq = getattr(self, "_call_seq", 0) + 1
            what = f"{what}#{self._call_seq:03d}"

        def run(stream):
            L.launch(what, fn, args, stream, flops, nbytes)
        return run


# ======================================================================================================
# ops
# ======================================================================================================
class EmbedGather:
    """model/layer.py:147-157 FeaturesEmbedding.forward(squeeze_dim=True)."""

    def __init__(self, plan, table, offsets_i32, n_fields, dim):
        self.table = table                  # Parameter [R, D]
        self.offsets = offsets_i32          # device int32 [F]
        self.F, self.D = n_fields, dim
        self.ids = torch.zeros((plan.B, n_fields), dtype=torch.int32, device=plan.device)
        self.idx = torch.empty((plan.B, n_fields), dtype=torch.int32, device=plan.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=plan.device)
        self.out = plan.new(n_fields * dim)
        plan.add(self)

    def build_fwd(self, plan):
        R = self.table.shape[0]
        oh, ldh = None, 0
        if plan.use_g2 and plan.shadow_wanted(self.out):      # the contractions read the embeddings through their bf16 shadow
            ptr, ldh = plan.shadow_view(self.out)
            oh = C.c_void_p(ptr)
            plan.mark_shadow(self.out)
        self.fwd_step = plan.call("cdc_embed_gather_fwd_h", _p(self.ids), _p(self.offsets), _p(self.table.data),
                                  self.out.cptr(), oh, C.c_int64(ldh), _p(self.idx), _p(self.err), C.c_int64(plan.B), self.F, self.D,
                                  C.c_int64(R), what="cdc_embed_gather_fwd")
        plan.fwd_steps.append(self.fwd_step)     # (a row-sharded trainer replaces this launch by its exchange)

    def build_bwd(self, plan, gs):
        # the gradient w.r.t. the gathered rows stays in self.out.grad; the table optimiser (optim.py) or the
        # drop-in dense-gradient path (functional.py) consumes it together with self.idx.
        if not gs.is_set(self.out):
            g = self.out.grad
            plan.bwd_steps.append(plan.call("cdc_fill_f32", g.cptr(), C.c_float(0.0), C.c_int64(g.rows * g.ld)))


class GLinear:
    """Several nn.Linear layers (optionally + ReLU + dropout) in one launch.

    groups: list of dicts {x: Buf, w: Parameter [N,K], b: Parameter [N] or None, act_cols: int or None}.
    Outputs are carved out of `out` buffers given per group (Buf [B, N]) or freshly allocated.
    """

    def __init__(self, plan, groups, relu=False, dropout=False, row_offsets=None, M=None):
        """a group with "half_only": True: its output feeds nothing but further bf16 contractions (the caller guarantees it); with
        the gemm2 path it and its gradient are kept as bf16 shadows only (no fp32 round trip through HBM)."""
        self.groups = groups
        self.relu = relu
        self.drop_p = plan.dropout if dropout else 0.0
        self.row_offsets = row_offsets
        self.M = plan.B if M is None else M
        self.seed = plan._next_seed()
        self.outs = []
        self.adopted = []                 # groups of ANOTHER launch whose grad-input this op's backward forms (adopt_bwd_x)
        for g in groups:
            N, K = g["w"].shape
            assert g["x"].cols == K, f"linear input has {g['x'].cols} columns, weight expects {K}"
            y = g.get("out")
            if y is None:
                y = plan.new(N, rows=g["x"].rows)
            assert y.cols == N
            act_cols = g.get("act_cols")
            act_cols = N if act_cols is None else act_cols
            g["act_cols"] = act_cols if (relu or self.drop_p > 0) else 0
            if g["act_cols"] > 0:
                y.mask = (plan.keep_scale if self.drop_p > 0 else 1.0, g["act_cols"])
            g["y"] = y
            self.outs.append(y)
        # bf16 operands from memory (csrc/gemm2.hip): plain weights, whole row range, 16-byte aligned column slices
        self.g2 = (plan.use_g2 and row_offsets is None and self.M > 0 and
                   all(isinstance(g["w"], (torch.Tensor, PView)) and not isinstance(g.get("b"), TView) for g in groups) and
                   all(g["x"].col0 % 8 == 0 and g["y"].col0 % 8 == 0 for g in groups))
        for g in groups:
            if plan.is_half_only(g["x"]) and not self.g2:
                raise RuntimeError("a linear layer outside the bf16-shadow path reads an activation that exists only as a shadow")
        if self.g2:
            for g in groups:
                plan.want_shadow(g["x"])                  # forward and grad-weight read x through its shadow
                plan.want_shadow(g["y"].grad)             # grad-input and grad-weight read dZ through its shadow
                if g.get("half_only"):
                    plan.want_shadow(g["y"])
                    plan.make_half_only(g["y"])
        # grad-weight split-K: enough (tile, row-slice) workgroups to fill 256 CUs several times over
        self.split_k = []
        for c0 in range(0, len(groups), L.MAX_GROUPS):
            chunk = groups[c0:c0 + L.MAX_GROUPS]
            tiles = sum(math.ceil(g["w"].shape[0] / 64) * math.ceil(g["w"].shape[1] / 64) for g in chunk)
            S = max(1, min(1024 // max(tiles, 1), max(self.M // 128, 1), 64))
            self.split_k.append(S)
            if S > 1:
                plan.need_gemm_ws(S * sum(g["w"].numel() + g["w"].shape[0] for g in chunk))
        plan.add(self)

    def adopt_bwd_x(self, other, groups):
        """`groups` of the launch `other` read an input that this launch's groups read too: their grad-input contributions become
        further reduction segments of THIS op's grad-input output instead of an output (and tiles) of their own in `other`'s.
        (PLE / MMoE: the gates read the level's input like the first expert layer, but their forward rides in the LAST expert
        layer's launch, which has tile slots to spare.)  Both launches must be on the bf16-shadow path."""
        if not (self.g2 and other.g2 and self.row_offsets is None and other.row_offsets is None and self.M == other.M):
            return False
        for g in groups:
            g["no_dx"] = True
            self.adopted.append(g)
        return True

    def _build_fwd_g2(self, plan):
        plan.ensure_shadows([g["x"] for g in self.groups], plan.fwd_steps)
        for c0 in range(0, len(self.groups), L.G2_MAX_OUT):
            chunk = self.groups[c0:c0 + L.G2_MAX_OUT]
            a = L.G2Args()
            a.n_out = a.n_seg = len(chunk)
            a.mode, a.relu, a.drop_p, a.mask_scale = 0, 1 if self.relu else 0, self.drop_p, 1.0
            a.seed = (self.seed + c0) & 0xFFFFFFFFFFFFFFFF
            a.seed_offset_dev = plan.step_dev.data_ptr()
            for i, g in enumerate(chunk):
                O, S = a.o[i], a.s[i]
                N, K = g["w"].shape
                y = g["y"]
                if plan.is_half_only(y):
                    O.y = None
                else:
                    O.y, O.ldy = y.ptr, y.ld
                if plan.shadow_wanted(y):                  # a later contraction reads y: its shadow comes out of this epilogue
                    O.yh, O.ldyh = plan.shadow_view(y)
                    plan.mark_shadow(y)
                else:
                    O.yh = None
                O.bias = None if g.get("b") is None else g["b"].data_ptr()
                O.mask, O.bn_partial = None, None
                O.M, O.N, O.act_cols, O.accumulate, O.mask_bf16, O.stream_id = self.M, N, g["act_cols"], 0, 0, i
                S.a, S.lda = plan.shadow_view(g["x"])
                wh, _ = plan.wshadow(g["w"])
                S.b, S.ldb = wh.data_ptr(), wh.stride(0)
                S.Kr, S.out = (K + 63) // 64 * 64, i
                plan._lin_producer[(id(y.root), y.col0, y.cols)] = {"op": self, "G": O, "step": len(plan.fwd_steps), "taken": False}
            self._keep = getattr(self, "_keep", []) + [a]
            fl = sum(2.0 * self.M * g["w"].shape[0] * g["w"].shape[1] for g in chunk)
            step = plan.call("cdc_gemm_bf16_nt", C.byref(a), what="cdc_glinear_fwd", flops=fl)
            self._fwd_calls = getattr(self, "_fwd_calls", []) + [step]
            plan.fwd_steps.append(step)

    def build_fwd(self, plan):
        if self.g2:
            return self._build_fwd_g2(plan)
        for c0 in range(0, len(self.groups), L.MAX_GROUPS):
            chunk = self.groups[c0:c0 + L.MAX_GROUPS]
            a = L.LinFwdArgs()
            a.n_groups = len(chunk)
            a.relu = 1 if self.relu else 0
            a.drop_p = self.drop_p
            a.seed = (self.seed + c0) & 0xFFFFFFFFFFFFFFFF
            a.seed_offset_dev = plan.step_dev.data_ptr()
            a.row_offsets = None if self.row_offsets is None else self.row_offsets.data_ptr() + 4 * c0
            for i, g in enumerate(chunk):
                G = a.g[i]
                N, K = g["w"].shape
                G.x, G.ldx = g["x"].ptr, g["x"].ld
                G.w, G.ldw = g["w"].data_ptr(), K
                G.bias = None if g.get("b") is None else g["b"].data_ptr()
                G.y, G.ldy = g["y"].ptr, g["y"].ld
                G.M, G.N, G.K = self.M, N, K
                G.act_cols = g["act_cols"]
                G.bn_partial = None
                # a BatchNorm that normalises this output next may ask the epilogue for its partial sums (BatchNorm.build_fwd)
                y = g["y"]
                plan._lin_producer[(id(y.root), y.col0, y.cols)] = {"op": self, "G": G, "step": len(plan.fwd_steps), "taken": False}
            self._keep = getattr(self, "_keep", []) + [a]
            fl = sum(2.0 * self.M * g["w"].shape[0] * g["w"].shape[1] for g in chunk)
            plan.fwd_steps.append(plan.call("cdc_glinear_fwd", C.byref(a), plan.prec, flops=fl))

    def build_bwd(self, plan, gs):
        # dZ of every group lives in y.grad (pre-activation gradient, see module docstring)
        for g in self.groups:
            if plan.is_half_only(g["y"]) or plan.is_half_only(g["y"].grad):
                if not plan.has_shadow(g["y"].grad):
                    raise RuntimeError("the gradient of a shadow-only activation was not written by a bf16 grad-input launch")
                continue
            plan.ensure_grad(g["y"], gs)
        if self.g2:                                  # grad-input and grad-weight read dZ through its bf16 shadow
            plan.ensure_shadows([g["y"].grad for g in self.groups + self.adopted], plan.bwd_steps)
        # ---- grad-weight / grad-bias
        # leaf weights (parameters): deferred to one launch with every other layer's at the end of backward.
        # fused weights (STAR's W_d*W_s: their gradient feeds a further backward op) and ragged groups: right here.
        defer = self.row_offsets is None and not any(isinstance(g["w"], TView) or isinstance(g.get("b"), TView) for g in self.groups)
        if defer:
            for g in self.groups:
                N, K = g["w"].shape
                dz = g["y"].grad
                gw = plan.param_grad(g["w"])
                acc_w = plan._claim_param(g["w"])
                db = None
                if g.get("b") is not None:
                    gb = plan.param_grad(g["b"])
                    acc_b = plan._claim_param(g["b"])
                    assert acc_b == acc_w
                    db = gb.data_ptr()
                plan._deferred_dw.append({"dz": dz.ptr, "lddz": dz.ld, "x": g["x"].ptr, "ldx": g["x"].ld, "dw": gw.data_ptr(), "lddw": K,
                                          "db": db, "M": self.M, "N": N, "K": K, "accumulate": 1 if acc_w else 0,
                                          "dzh": plan.shadow_view(dz) if self.g2 else None,
                                          "xh": plan.shadow_view(g["x"]) if self.g2 else None,
                                          "needs_shadows": plan.is_half_only(g["x"]) or plan.is_half_only(g["y"]) or plan.is_half_only(dz)})
        for c0 in (range(0, len(self.groups), L.MAX_GROUPS) if not defer else []):
            chunk = self.groups[c0:c0 + L.MAX_GROUPS]
            a = L.LinBwdwArgs()
            a.n_groups = len(chunk)
            a.split_k = self.split_k[c0 // L.MAX_GROUPS]
            plan._gemm_ws_users.append(a)          # .workspace is set when the plan is finalised (the buffer may still grow)
            a.row_offsets = None if self.row_offsets is None else self.row_offsets.data_ptr() + 4 * c0
            for i, g in enumerate(chunk):
                G = a.g[i]
                N, K = g["w"].shape
                dz = g["y"].grad
                G.dz, G.lddz = dz.ptr, dz.ld
                G.x, G.ldx = g["x"].ptr, g["x"].ld
                gw = plan.param_grad(g["w"])
                acc_w = plan._claim_param(g["w"])
                G.dw, G.lddw = gw.data_ptr(), K
                if g.get("b") is not None:
                    gb = plan.param_grad(g["b"])
                    acc_b = plan._claim_param(g["b"])
                    assert acc_b == acc_w
                    G.db = gb.data_ptr()
                else:
                    G.db = None
                G.M, G.N, G.K = self.M, N, K
                G.accumulate = 1 if acc_w else 0
            self._keep.append(a)
            fl = sum(2.0 * self.M * g["w"].shape[0] * g["w"].shape[1] for g in chunk)
            plan.bwd_steps.append(plan.call("cdc_glinear_bwd_w", C.byref(a), plan.prec, flops=fl))
        if getattr(self, "skip_bwd_x", False):       # the grad-input of this launch is formed elsewhere (CGCMid)
            return
        # ---- grad-input: groups reading the same x reduce into one output
        outs = []      # list of (x Buf, [group indices]); indices run over this op's groups, then the adopted ones
        allg = self.groups + self.adopted
        for gi, g in enumerate(allg):
            if g.get("no_dx") and gi < len(self.groups):
                continue
            x = g["x"]
            for o in outs:
                if o[0].root is x.root and o[0].col0 == x.col0 and o[0].cols == x.cols:
                    o[1].append(gi)
                    break
            else:
                outs.append((x, [gi]))
        # pack outputs into launches of <= MAX_GROUPS outputs / segments
        launches, cur_o, cur_s = [], [], 0
        for o in outs:
            if len(o[1]) > L.MAX_GROUPS:
                raise RuntimeError("more than %d linear groups share one input" % L.MAX_GROUPS)
            if cur_o and (len(cur_o) + 1 > L.MAX_GROUPS or cur_s + len(o[1]) > L.MAX_GROUPS):
                launches.append(cur_o)
                cur_o, cur_s = [], 0
            cur_o.append(o)
            cur_s += len(o[1])
        if cur_o:
            launches.append(cur_o)
        if self.row_offsets is not None:
            # ragged rows: group g owns rows [ro[g], ro[g+1]) of the (shared) buffers: one output per group, in group
            # order, and ONE accumulate decision per distinct input buffer
            outs = [(g["x"], [gi]) for gi, g in enumerate(self.groups) if not g.get("no_dx")]
            assert len(outs) == len(self.groups) and len(outs) <= L.MAX_GROUPS, "ragged launch limited to one chunk"
            launches = [outs]
        if self.g2:
            return self._build_bwd_x_g2(plan, gs, outs, allg)
        ragged_acc = {}
        for la in launches:
            # all outputs of one launch share the mask scale (one plan-wide dropout rate)
            a = L.LinBwdxArgs()
            a.n_out = len(la)
            a.mask_scale = 1.0
            si = 0
            first_group = la[0][1][0]
            a.row_offsets = None if self.row_offsets is None else self.row_offsets.data_ptr() + 4 * first_group
            for oi, (x, gis) in enumerate(la):
                O = a.o[oi]
                xg = x.grad
                O.dx, O.lddx = xg.ptr, xg.ld
                if x.mask is not None:
                    O.mask_y, O.ldmask = x.ptr, x.ld
                    O.mask_cols = min(x.mask[1], x.cols)
                    a.mask_scale = x.mask[0]
                else:
                    O.mask_y = None
                    O.mask_cols = 0
                O.M, O.K = self.M, x.cols
                if self.row_offsets is not None:
                    key = (id(x.root), x.col0, x.cols)
                    if key not in ragged_acc:
                        ragged_acc[key] = gs.claim(x)
                    O.accumulate = 1 if ragged_acc[key] else 0
                else:
                    O.accumulate = 1 if gs.claim(x) else 0
                for gi in gis:
                    g = self.groups[gi]
                    S = a.s[si]
                    dz = g["y"].grad
                    S.dz, S.lddz = dz.ptr, dz.ld
                    S.w, S.ldw = g["w"].data_ptr(), g["w"].shape[1]
                    half = plan.prec == L.PREC_BF16 and g["w"].shape[0] % 8 == 0 and dz.ptr % 16 == 0 and dz.ld % 4 == 0
                    wt = plan.wt_of(g["w"], half=half)
                    S.wt, S.ldwt = wt.data_ptr(), g["w"].shape[0]
                    S.wt_bf16 = 1 if half else 0
                    S.N = g["w"].shape[0]
                    S.out = oi
                    si += 1
            a.n_seg = si
            self._keep.append(a)
            fl = sum(2.0 * self.M * self.groups[gi]["w"].shape[0] * self.groups[gi]["w"].shape[1] for _, gis in la for gi in gis)
            plan.bwd_steps.append(plan.call("cdc_glinear_bwd_x", C.byref(a), plan.prec, flops=fl))


    def _build_bwd_x_g2(self, plan, gs, outs, allg):
        """grad-input from the bf16 shadows of dZ and the per-step W^T copies; outputs: x.grad in fp32, plus its bf16 shadow
        when the layer that produced x reads its dZ through one"""
        launches, cur_o, cur_s = [], [], 0
        for o in outs:
            if len(o[1]) > L.G2_MAX_SEG:
                raise RuntimeError("more than %d linear groups share one input" % L.G2_MAX_SEG)
            if cur_o and (len(cur_o) + 1 > L.G2_MAX_OUT or cur_s + len(o[1]) > L.G2_MAX_SEG):
                launches.append(cur_o)
                cur_o, cur_s = [], 0
            cur_o.append(o)
            cur_s += len(o[1])
        if cur_o:
            launches.append(cur_o)
        for la in launches:
            a = L.G2Args()
            a.n_out = len(la)
            a.mode, a.relu, a.drop_p, a.mask_scale, a.seed, a.seed_offset_dev = 1, 0, 0.0, 1.0, 0, None
            si = 0
            for oi, (x, gis) in enumerate(la):
                O = a.o[oi]
                xg = x.grad
                acc = gs.claim(x)                          # (invalidates an older shadow of x.grad)
                half = plan.is_half_only(xg)               # (the values of x may be bf16-only while their gradient is not)
                if half:
                    if acc:
                        raise RuntimeError("an activation kept as a bf16 shadow only feeds more than one grad-input launch")
                    O.y = None
                else:
                    O.y, O.ldy = xg.ptr, xg.ld
                if half or plan.shadow_wanted(xg):
                    O.yh, O.ldyh = plan.shadow_view(xg)
                    plan.mark_shadow(xg)
                else:
                    O.yh = None
                O.bias, O.bn_partial = None, None
                if x.mask is not None:
                    if plan.has_shadow(x):                 # the sign of the activation is the same in its bf16 shadow: half the bytes
                        (O.mask, O.ldmask), O.mask_bf16 = plan.shadow_view(x), 1
                    else:
                        O.mask, O.ldmask, O.mask_bf16 = x.ptr, x.ld, 0
                    O.act_cols = min(x.mask[1], x.cols)
                    a.mask_scale = x.mask[0]
                else:
                    O.mask, O.act_cols, O.mask_bf16 = None, 0, 0
                O.M, O.N, O.accumulate, O.stream_id = self.M, x.cols, 1 if acc else 0, oi
                for gi in gis:
                    g = allg[gi]
                    S = a.s[si]
                    S.a, S.lda = plan.shadow_view(g["y"].grad)
                    _, wt = plan.wshadow(g["w"])
                    S.b, S.ldb = wt.data_ptr(), wt.stride(0)
                    S.Kr, S.out = (g["w"].shape[0] + 63) // 64 * 64, oi
                    si += 1
            a.n_seg = si
            self._keep.append(a)
            fl = sum(2.0 * self.M * allg[gi]["w"].shape[0] * allg[gi]["w"].shape[1] for _, gis in la for gi in gis)
            plan.bwd_steps.append(plan.call("cdc_gemm_bf16_nt", C.byref(a), what="cdc_glinear_bwd_x", flops=fl))


class GatePool:
    """softmax gates + weighted expert pooling (model/ple.py:105-123, model/mmoe.py:58-60).

    experts: Buf [B, n_expert*H]; gates: list of (logits Buf [B, n_sel], sel list[int]).
    """

    def __init__(self, plan, experts, n_expert, H, gates, outs=None):
        assert experts.cols == n_expert * H
        self.experts, self.n_expert, self.H = experts, n_expert, H
        self.gates = gates
        self.outs = outs if outs is not None else [plan.new(H) for _ in gates]
        self.probs = [torch.empty((plan.B, len(sel)), dtype=torch.float32, device=plan.device) for _, sel in gates]
        for _, sel in gates:
            if len(sel) > L.MAX_SEL:
                raise RuntimeError(f"a gate mixes {len(sel)} experts; the pooling kernel supports {L.MAX_SEL}")
        self._keep = []
        plan.add(self)

    def build_fwd(self, plan):
        for c0 in range(0, len(self.gates), L.MAX_GATES):
            a = L.PoolFwdArgs()
            chunk = self.gates[c0:c0 + L.MAX_GATES]
            a.n_gates, a.n_expert, a.H, a.B = len(chunk), self.n_expert, self.H, plan.B
            a.experts, a.ld_exp = self.experts.ptr, self.experts.ld
            for i, (lg, sel) in enumerate(chunk):
                G = a.gate[i]
                G.logits, G.ld_logits = lg.ptr, lg.ld
                G.out, G.ld_out = self.outs[c0 + i].ptr, self.outs[c0 + i].ld
                if plan.use_g2 and plan.shadow_wanted(self.outs[c0 + i]):
                    G.out_h, G.ld_out_h = plan.shadow_view(self.outs[c0 + i])
                    plan.mark_shadow(self.outs[c0 + i])
                else:
                    G.out_h = None
                G.probs = self.probs[c0 + i].data_ptr()
                G.n_sel = len(sel)
                for j, e in enumerate(sel):
                    G.sel[j] = e
            self._keep.append(a)
            plan.fwd_steps.append(plan.call("cdc_gate_pool_fwd", C.byref(a)))

    def build_bwd(self, plan, gs):
        acc = gs.claim(self.experts)
        for c0 in range(0, len(self.gates), L.MAX_GATES):
            a = L.PoolBwdArgs()
            chunk = self.gates[c0:c0 + L.MAX_GATES]
            a.n_gates, a.n_expert, a.H, a.B = len(chunk), self.n_expert, self.H, plan.B
            a.experts, a.ld_exp = self.experts.ptr, self.experts.ld
            eg = self.experts.grad
            a.d_experts, a.ld_dexp = eg.ptr, eg.ld
            if self.experts.mask is not None:
                assert self.experts.mask[1] >= self.experts.cols, "expert buffer is only partly activation-masked"
                a.mask_relu, a.mask_scale = 1, self.experts.mask[0]
            else:
                a.mask_relu, a.mask_scale = 0, 1.0
            a.accumulate = 1 if (acc or c0 > 0) else 0
            if plan.use_g2 and plan.shadow_wanted(eg):       # the experts' grad-input / grad-weight read dZ through its shadow
                a.d_experts_h, a.ld_dexp_h = plan.shadow_view(eg)
                plan.mark_shadow(eg)
            else:
                a.d_experts_h = None
            for i, (lg, sel) in enumerate(chunk):
                G = a.gate[i]
                plan.ensure_grad(self.outs[c0 + i], gs)
                og = self.outs[c0 + i].grad
                G.d_out, G.ld_dout = og.ptr, og.ld
                G.probs = self.probs[c0 + i].data_ptr()
                lgg = lg.grad
                if gs.claim(lg):
                    raise RuntimeError("gate logits feed more than one consumer")
                G.d_logits, G.ld_dlogits = lgg.ptr, lgg.ld
                if plan.use_g2 and plan.shadow_wanted(lgg):
                    G.d_logits_h, G.ld_dlogits_h = plan.shadow_view(lgg)
                    plan.mark_shadow(lgg)
                else:
                    G.d_logits_h = None
                G.n_sel = len(sel)
                for j, e in enumerate(sel):
                    G.sel[j] = e
            self._keep.append(a)
            plan.bwd_steps.append(plan.call("cdc_gate_pool_bwd", C.byref(a)))


class CGCMid:
    """The boundary between two extraction levels of PLE as ONE launch per direction (csrc/cgc.hip; model/ple.py:54-57,96-125):
    GatePool(level k) -> GLinear(level k+1: single-layer experts + gates) -> GatePool(level k+1).  Built from the three ops the
    model described (it takes their place in plan.ops and uses their buffers): the pooled level-k vectors exist only as bf16
    shadows (the contraction operand and the grad-weight operand), their gradient never leaves the backward launch, and the dZ of
    both expert levels is written as bf16 alone."""

    H1, H2 = 128, 64                      # the widths csrc/cgc.hip is instantiated for (config.py:39-42: ((256,128),(64,)))
    enabled = True                        # tests set this to False to build the three launches per direction at a matching shape
    MAX_B = 8192                          # batch rows up to which the fused boundary beats the three launches (see match)

    @staticmethod
    def _same(a, b):
        return a.root is b.root and a.col0 == b.col0 and a.cols == b.cols and a.rows == b.rows

    @classmethod
    def match(cls, plan, p1, lin, p2):
        if not (isinstance(p1, GatePool) and isinstance(lin, GLinear) and isinstance(p2, GatePool)):
            return None
        if not (plan.use_g2 and lin.g2 and lin.row_offsets is None and lin.M == plan.B and not lin.adopted and lin.relu):
            return None
        if (p1.H, p2.H) != (cls.H1, cls.H2) or len(lin.groups) > L.G2_MAX_OUT:
            return None
        # one workgroup (16 rows) per CU at a time: the fused launches scale linearly with the batch from B = 4096 on, the three
        # launches they replace are still latency-bound there.  Measured stand-alone, fused vs unfused, forward + backward: 56 vs 80 us
        # at B = 4096, 200 vs 188 at 16384, 396 vs 385 at 32768 (profiles/round3/asymptote.txt): fused up to 8192 rows per GPU
        if plan.B > cls.MAX_B:
            return None
        ne2, ng2 = p2.n_expert, len(p2.gates)
        if (len(lin.groups) != ne2 + ng2 or p1.n_expert > L.MID_MAX_EXPERT or ne2 > L.MID_MAX_EXPERT or
                len(p1.gates) > L.MID_MAX_GATE or ng2 > L.MID_MAX_GATE):
            return None
        for _, sel in list(p1.gates) + list(p2.gates):
            if list(sel) != sorted(set(sel)):
                return None
        # the fused launches keep a workgroup's expert rows, pooled vectors and gradients in LDS: more domains than that holds
        # (e.g. PLE with 6 domains x 2 specific + 2 shared experts: 160 KB) keep the three launches per direction
        if plan.lib.cdc_cgc_mid_fits(p1.n_expert, len(p1.gates), ne2, ng2) != 1:
            return None

        def src_of(x):
            for i, o in enumerate(p1.outs):
                if cls._same(o, x):
                    return i
            return None
        srcs = []
        for e, g in enumerate(lin.groups[:ne2]):
            y = g["y"]
            if not (tuple(g["w"].shape) == (cls.H2, cls.H1) and g["act_cols"] == cls.H2 and y.root is p2.experts.root and
                    y.col0 == p2.experts.col0 + e * cls.H2 and isinstance(g["w"], torch.Tensor)):
                return None
            srcs.append(src_of(g["x"]))
        for t, g in enumerate(lin.groups[ne2:]):
            lg, sel = p2.gates[t]
            if not (cls._same(g["y"], lg) and g["act_cols"] == 0 and g["w"].shape[0] == len(sel) and g["w"].shape[1] == cls.H1 and
                    isinstance(g["w"], torch.Tensor)):
                return None
            srcs.append(src_of(g["x"]))
        if any(s_ is None for s_ in srcs):
            return None
        for s_ in range(len(p1.outs)):                     # K steps of 32 per grad-input tile (csrc/cgc.hip MID_MAXSTEP)
            if sum(cls.H2 // 32 for x in srcs[:ne2] if x == s_) + sum(1 for x in srcs[ne2:] if x == s_) > 6:
                return None
        if plan.is_half_only(p1.experts) or plan.is_half_only(p2.experts):
            return None
        return srcs

    def __init__(self, plan, p1, lin, p2, srcs):
        self.p1, self.lin, self.p2, self.srcs = p1, lin, p2, srcs
        i = plan.ops.index(p1)
        assert plan.ops[i + 1] is lin and plan.ops[i + 2] is p2
        plan.ops[i:i + 3] = [self]
        for o in p1.outs:                                  # never written as fp32: a stray reader shows up as NaN
            plan.make_value_half_only(o)
        # dZ of both expert levels: bf16 alone (read by the grad-weight launches and the level-k grad-input launch)
        plan.make_grad_half_only(p1.experts)
        plan.make_grad_half_only(p2.experts)
        self._keep = []

    def build_fwd(self, plan):
        p1, lin, p2 = self.p1, self.lin, self.p2
        ne2 = p2.n_expert
        a = L.CgcMidFwdArgs()
        a.B, a.H1, a.H2 = plan.B, self.H1, self.H2
        a.n_exp1, a.n_gate1, a.n_exp2, a.n_gate2 = p1.n_expert, len(p1.gates), ne2, len(p2.gates)
        a.ex1, a.ld_ex1 = p1.experts.ptr, p1.experts.ld
        a.ex2, a.ld_ex2 = p2.experts.ptr, p2.experts.ld
        a.relu, a.drop_p, a.seed, a.seed_offset_dev = 1 if lin.relu else 0, lin.drop_p, lin.seed & 0xFFFFFFFFFFFFFFFF, plan.step_dev.data_ptr()
        for i, (lg, sel) in enumerate(p1.gates):
            G = a.g1[i]
            G.logits, G.ld_logits = lg.ptr, lg.ld
            G.probs = p1.probs[i].data_ptr()
            G.pooled_h, G.ld_pooled_h = plan.shadow_view(p1.outs[i])
            plan.mark_shadow(p1.outs[i])
            G.n_sel = len(sel)
            for j, e in enumerate(sel):
                G.sel[j] = e
        for e, g in enumerate(lin.groups[:ne2]):
            E = a.e2[e]
            wh, _ = plan.wshadow(g["w"])
            E.w, E.ldw = wh.data_ptr(), wh.stride(0)
            E.bias = None if g.get("b") is None else g["b"].data_ptr()
            E.src, E.stream_id = self.srcs[e], e           # stream_id: the group's index in the unfused launch
        for t, g in enumerate(lin.groups[ne2:]):
            G = a.g2[t]
            lg, sel = p2.gates[t]
            wh, _ = plan.wshadow(g["w"])
            G.w, G.ldw = wh.data_ptr(), wh.stride(0)
            G.bias = None if g.get("b") is None else g["b"].data_ptr()
            G.probs = p2.probs[t].data_ptr()
            o = p2.outs[t]
            G.out, G.ld_out = o.ptr, o.ld
            if plan.shadow_wanted(o):
                G.out_h, G.ld_out_h = plan.shadow_view(o)
                plan.mark_shadow(o)
            else:
                G.out_h = None
            G.src, G.n_sel = self.srcs[ne2 + t], len(sel)
            for j, e in enumerate(sel):
                G.sel[j] = e
        self._keep.append(a)
        fl = sum(2.0 * plan.B * g["w"].shape[0] * g["w"].shape[1] for g in lin.groups)
        plan.fwd_steps.append(plan.call("cdc_cgc_mid_fwd", C.byref(a), flops=fl))

    def build_bwd(self, plan, gs):
        p1, lin, p2 = self.p1, self.lin, self.p2
        ne2 = p2.n_expert
        if gs.claim(p1.experts) or gs.claim(p2.experts):
            raise RuntimeError("CGCMid: an expert buffer's gradient has another writer")
        a = L.CgcMidBwdArgs()
        a.B, a.H1, a.H2 = plan.B, self.H1, self.H2
        a.n_exp1, a.n_gate1, a.n_exp2, a.n_gate2 = p1.n_expert, len(p1.gates), ne2, len(p2.gates)
        a.ex1, a.ld_ex1 = p1.experts.ptr, p1.experts.ld
        a.ex2, a.ld_ex2 = p2.experts.ptr, p2.experts.ld
        a.dz1_h, a.ld_dz1_h = plan.shadow_view(p1.experts.grad)
        a.dz2_h, a.ld_dz2_h = plan.shadow_view(p2.experts.grad)
        plan.mark_shadow(p1.experts.grad)
        plan.mark_shadow(p2.experts.grad)
        for ex, mk, sc in ((p1.experts, "mask1", "scale1"), (p2.experts, "mask2", "scale2")):
            if ex.mask is not None:
                assert ex.mask[1] >= ex.cols, "expert buffer is only partly activation-masked"
                setattr(a, mk, 1)
                setattr(a, sc, ex.mask[0])
            else:
                setattr(a, mk, 0)
                setattr(a, sc, 1.0)

        def fill_logit_grads(G, lg):
            lgg = lg.grad
            if gs.claim(lg):
                raise RuntimeError("gate logits feed more than one consumer")
            G.d_logits, G.ld_dlogits = lgg.ptr, lgg.ld
            if plan.shadow_wanted(lgg):
                G.d_logits_h, G.ld_dlogits_h = plan.shadow_view(lgg)
                plan.mark_shadow(lgg)
            else:
                G.d_logits_h = None
        for i, (lg, sel) in enumerate(p1.gates):
            G = a.g1[i]
            G.probs = p1.probs[i].data_ptr()
            fill_logit_grads(G, lg)
            G.n_sel = len(sel)
            for j, e in enumerate(sel):
                G.sel[j] = e
        for e, g in enumerate(lin.groups[:ne2]):
            _, wt = plan.wshadow(g["w"])
            a.e2[e].wt, a.e2[e].ldwt, a.e2[e].src = wt.data_ptr(), wt.stride(0), self.srcs[e]
        for t, g in enumerate(lin.groups[ne2:]):
            G = a.g2[t]
            lg, sel = p2.gates[t]
            plan.ensure_grad(p2.outs[t], gs)
            og = p2.outs[t].grad
            G.d_out, G.ld_dout = og.ptr, og.ld
            G.probs = p2.probs[t].data_ptr()
            fill_logit_grads(G, lg)
            _, wt = plan.wshadow(g["w"])
            G.wt, G.ldwt = wt.data_ptr(), wt.stride(0)
            G.src, G.n_sel = self.srcs[ne2 + t], len(sel)
            for j, e in enumerate(sel):
                G.sel[j] = e
        self._keep.append(a)
        fl = sum(2.0 * plan.B * g["w"].shape[0] * g["w"].shape[1] for g in lin.groups)
        plan.bwd_steps.append(plan.call("cdc_cgc_mid_bwd", C.byref(a), flops=fl))
        # the level-k+1 grad-weight contractions stay with the batched launches at the end of backward
        lin.skip_bwd_x = True
        lin.build_bwd(plan, gs)


def fuse_cgc_mid(plan):
    """Replaces every (GatePool, GLinear, GatePool) run of plan.ops that is a PLE level boundary of the instantiated shape by ONE
    CGCMid op.  Called by the model that KNOWS the pooled level-k vectors have no other reader (model/ple.py: ple_inputs)."""
    if not CGCMid.enabled:
        return 0
    n = 0
    i = 0
    while i + 2 < len(plan.ops):
        srcs = CGCMid.match(plan, plan.ops[i], plan.ops[i + 1], plan.ops[i + 2])
        if srcs is not None:
            CGCMid(plan, plan.ops[i], plan.ops[i + 1], plan.ops[i + 2], srcs)
            n += 1
        i += 1
    return n


class ExpertPair:
    """Two consecutive BatchNorm-free expert layers — GLinear(x -> H1, ReLU, dropout; hidden kept as a bf16 shadow only) and
    GLinear(H1 -> H2, ReLU, dropout [+ gate projections of x riding along]) — as ONE forward launch (csrc/pair.hip; model/ple.py:83-94,
    model/layer.py:185-196).  Built from the two ops the model described (it takes their place in plan.ops and uses their buffers);
    the forward is bit-identical to their two launches, the backward IS theirs."""

    H1, H2 = 256, 128                     # the widths csrc/pair.hip is instantiated for (config.py:39-42: ((256,128),(64,)))
    enabled = True                        # tests set this to False to build the two launches at a matching shape

    @classmethod
    def match(cls, plan, la, lb):
        if not (isinstance(la, GLinear) and isinstance(lb, GLinear) and plan.use_g2 and la.g2 and lb.g2):
            return False
        if not (la.row_offsets is None and lb.row_offsets is None and la.M == plan.B and lb.M == plan.B and la.relu and lb.relu and
                la.drop_p == lb.drop_p and plan.B < (1 << 31)):
            return False
        n = len(la.groups)
        if not (0 < n <= L.PAIR_MAX_EXPERT and n <= len(lb.groups) <= 2 * n and len(lb.groups) <= L.G2_MAX_OUT):
            return False
        K = la.groups[0]["w"].shape[1]
        for i in range(n):
            ga, gb = la.groups[i], lb.groups[i]
            if not (isinstance(ga["w"], torch.Tensor) and isinstance(gb["w"], torch.Tensor) and tuple(ga["w"].shape) == (cls.H1, K) and
                    tuple(gb["w"].shape) == (cls.H2, cls.H1) and ga["act_cols"] == cls.H1 and gb["act_cols"] == cls.H2):
                return False
            if not (CGCMid._same(ga["y"], gb["x"]) and plan.is_half_only(ga["y"]) and not plan.is_half_only(ga["x"])):
                return False
        for j, g in enumerate(lb.groups[n:]):                 # the riders: narrow, no activation, reading what expert j reads
            if not (isinstance(g["w"], torch.Tensor) and g["w"].shape[0] <= 16 and g["w"].shape[1] == K and g["act_cols"] == 0 and
                    CGCMid._same(g["x"], la.groups[j]["x"]) and not plan.is_half_only(g["y"])):
                return False
        return True

    def __init__(self, plan, la, lb):
        self.la, self.lb = la, lb
        i = plan.ops.index(la)
        assert plan.ops[i + 1] is lb
        plan.ops[i:i + 2] = [self]
        self._keep = []
        for op in (la, lb):                                    # (their backward builders append argument blocks here)
            op._keep = getattr(op, "_keep", [])

    def build_fwd(self, plan):
        la, lb = self.la, self.lb
        n = len(la.groups)
        plan.ensure_shadows([g["x"] for g in la.groups], plan.fwd_steps)
        a = L.ExpertPairArgs()
        K = la.groups[0]["w"].shape[1]
        a.n_expert, a.M, a.K1r, a.H1, a.H2 = n, plan.B, (K + 63) // 64 * 64, self.H1, self.H2
        a.relu, a.drop_p = 1, la.drop_p
        a.seed1, a.seed2 = la.seed & 0xFFFFFFFFFFFFFFFF, lb.seed & 0xFFFFFFFFFFFFFFFF
        a.seed_offset_dev = plan.step_dev.data_ptr()
        for i in range(n):
            ga, gb, E = la.groups[i], lb.groups[i], a.e[i]
            E.x, E.ldx = plan.shadow_view(ga["x"])
            w1, _ = plan.wshadow(ga["w"])
            w2, _ = plan.wshadow(gb["w"])
            E.w1, E.ldw1 = w1.data_ptr(), w1.stride(0)
            E.w2, E.ldw2 = w2.data_ptr(), w2.stride(0)
            E.b1 = None if ga.get("b") is None else ga["b"].data_ptr()
            E.b2 = None if gb.get("b") is None else gb["b"].data_ptr()
            E.h, E.ldh = plan.shadow_view(ga["y"])
            plan.mark_shadow(ga["y"])
            y = gb["y"]
            if plan.is_half_only(y):
                E.y = None
            else:
                E.y, E.ldy = y.ptr, y.ld
            if plan.shadow_wanted(y):
                E.yh, E.ldyh = plan.shadow_view(y)
                plan.mark_shadow(y)
            else:
                E.yh = None
            E.stream1, E.stream2 = i, i                        # the group's index in each of the two launches
            E.ws, E.ns = None, 0
        for j, g in enumerate(lb.groups[n:]):
            E = a.e[j]
            ws, _ = plan.wshadow(g["w"])
            E.ws, E.ldws = ws.data_ptr(), ws.stride(0)
            E.bs = None if g.get("b") is None else g["b"].data_ptr()
            E.ys, E.ldys, E.ns = g["y"].ptr, g["y"].ld, g["w"].shape[0]
            if plan.shadow_wanted(g["y"]):
                raise RuntimeError("ExpertPair: a rider's output is read by a bf16 contraction (no shadow is written for it)")
        self._keep.append(a)
        fl = sum(2.0 * plan.B * g["w"].shape[0] * g["w"].shape[1] for g in la.groups + lb.groups)
        plan.fwd_steps.append(plan.call("cdc_expert_pair_fwd", C.byref(a), what="cdc_glinear_pair_fwd", flops=fl))

    def build_bwd(self, plan, gs):
        self.lb.build_bwd(plan, gs)
        self.la.build_bwd(plan, gs)


def fuse_expert_pair(plan):
    """Replaces every (GLinear, GLinear) run of plan.ops that is the two layers of a level's experts in the instantiated shape by ONE
    ExpertPair op."""
    if not ExpertPair.enabled:
        return 0
    n = 0
    i = 0
    while i + 1 < len(plan.ops):
        if ExpertPair.match(plan, plan.ops[i], plan.ops[i + 1]):
            ExpertPair(plan, plan.ops[i], plan.ops[i + 1])
            n += 1
        i += 1
    return n


class BatchNorm:
    """BatchNorm1d (+ReLU, +dropout) over column segments sharing one launch.

    segs: list of dicts {x: Buf, bn: module-like with weight/bias/running_mean/running_var/num_batches_tracked
          (or explicit tensors gamma/beta), out: Buf or None, row_group: int}
    """

    def __init__(self, plan, segs, relu=True, dropout=True, eps=1e-5, momentum=0.1, row_offsets=None, M=None, skip_le1=False):
        self.segs = segs
        self.skip_le1 = skip_le1
        self.relu = relu
        self.drop_p = plan.dropout if dropout else 0.0
        self.eps, self.momentum = eps, momentum
        self.row_offsets = row_offsets
        self.M = plan.B if M is None else M
        self.seed = plan._next_seed()
        self.outs = []
        total_c = 0
        for s in segs:
            Cn = s["x"].cols
            y = s.get("out")
            if y is None:
                y = plan.new(Cn, rows=s["x"].rows)
            s["y"] = y
            s["save_mean"] = torch.zeros(Cn, dtype=torch.float32, device=plan.device)
            s["save_invstd"] = torch.ones(Cn, dtype=torch.float32, device=plan.device)
            self.outs.append(y)
            total_c += Cn
        self._keep = []
        for c0 in range(0, len(segs), L.MAX_BN_SEGS):
            plan.need_bn_ws(self.M, sum(s["x"].cols for s in segs[c0:c0 + L.MAX_BN_SEGS]))
        plan.add(self)

    def build_fwd(self, plan):
        for c0 in range(0, len(self.segs), L.MAX_BN_SEGS):
            chunk = self.segs[c0:c0 + L.MAX_BN_SEGS]
            a = L.BnFwdArgs()
            a.n_seg, a.training, a.relu = len(chunk), 1 if plan.training else 0, 1 if self.relu else 0
            a.skip_le1 = 1 if self.skip_le1 else 0
            a.eps, a.momentum, a.drop_p = self.eps, self.momentum, self.drop_p
            a.seed, a.seed_offset_dev = (self.seed + c0) & 0xFFFFFFFFFFFFFFFF, plan.step_dev.data_ptr()
            a.M = self.M
            a.row_offsets = None if self.row_offsets is None else self.row_offsets.data_ptr()
            a.workspace = plan._bn_ws.data_ptr()
            for i, s in enumerate(chunk):
                S = a.s[i]
                S.x, S.ldx = s["x"].ptr, s["x"].ld
                S.y, S.ldy = s["y"].ptr, s["y"].ld
                S.half = 0
                prod = plan._lin_producer.get((id(s["x"].root), s["x"].col0, s["x"].cols))
                s["_lin_g2"] = bool(prod is not None and getattr(prod["op"], "g2", False) and prod["G"].act_cols == 0)
                # the caller vouches (seg["half_only"]) that y is read by bf16 contractions only — and one did ask for the shadow:
                # y then exists as bf16 alone (its sign is the relu/dropout mask of the backward)
                if s.get("half_only") and plan.use_g2 and plan.shadow_wanted(s["y"]) and self.row_offsets is None:
                    plan.make_value_half_only(s["y"])
                    S.y = None
                S.gamma, S.beta = s["gamma"].data_ptr(), s["beta"].data_ptr()
                S.running_mean, S.running_var = s["running_mean"].data_ptr(), s["running_var"].data_ptr()
                S.save_mean, S.save_invstd = s["save_mean"].data_ptr(), s["save_invstd"].data_ptr()
                nbt = s.get("num_batches_tracked")
                S.num_batches_tracked = None if nbt is None else nbt.data_ptr()
                S.C = s["x"].cols
                S.row_group = s.get("row_group", 0)
                if plan.use_g2 and plan.shadow_wanted(s["y"]):
                    S.yh, S.ldyh = plan.shadow_view(s["y"])
                    plan.mark_shadow(s["y"])
                else:
                    S.yh = None
            self._fuse_stats(plan, a, chunk)
            self._keep.append(a)
            if plan.dist is not None and plan.training:
                # global-batch statistics: partial sums -> all-reduce -> normalise
                total_c = sum(s["x"].cols for s in chunk)
                ex = torch.zeros(2 * total_c + len(chunk), dtype=torch.float64, device=plan.device)
                a.phase, a.exchange = 1, ex.data_ptr()
                a2 = L.BnFwdArgs.from_buffer_copy(a)
                a2.phase = 2
                self._keep += [ex, a2]
                plan.fwd_steps.append(plan.call("cdc_bn_fwd", C.byref(a), what="cdc_bn_fwd(stats)"))
                plan.fwd_steps.append(plan.comm(lambda ex=ex: plan.dist.all_reduce_sum(ex)))
                plan.fwd_steps.append(plan.call("cdc_bn_fwd", C.byref(a2), what="cdc_bn_fwd(apply)"))
            else:
                step = plan.call("cdc_bn_fwd", C.byref(a))
                self._fwd_calls = getattr(self, "_fwd_calls", []) + [step]
                plan.fwd_steps.append(step)
            plan._last_bn_step = len(plan.fwd_steps)

    def _fuse_stats(self, plan, a, chunk):
        """If every segment of this launch normalises the output of a plain linear launch (no activation, same rows, no
        BatchNorm launch in between — the partial-sum workspace is shared), those launches' epilogues write the partial
        sums and this launch skips its statistics pass (one read of the activations less per BatchNorm)."""
        a.stats_ready = 0
        if not plan.training or self.row_offsets is not None:
            return
        prods = []
        for s in chunk:
            x = s["x"]
            p = plan._lin_producer.get((id(x.root), x.col0, x.cols))
            if (p is None or p["taken"] or p["op"].row_offsets is not None or p["op"].M != self.M or p["G"].act_cols != 0 or
                    p["step"] < plan._last_bn_step):
                return
            prods.append(p)
        total_c = sum(s["x"].cols for s in chunk)
        col0 = 0
        for s, p in zip(chunk, prods):
            p["taken"] = True
            p["G"].bn_partial, p["G"].bn_col0, p["G"].bn_total_c = plan._bn_ws.data_ptr(), col0, total_c
            col0 += s["x"].cols
        a.stats_ready = 1

    def build_bwd(self, plan, gs):
        # segments that normalise the SAME input over the same rows (STAR all-towers: every domain_norm reads the
        # embeddings) cannot write one gradient buffer concurrently: each writes its own slice, summed afterwards
        keys = [(id(s["x"].root), s["x"].col0, s["x"].cols) for s in self.segs]
        shared = self.row_offsets is None and len(set(keys)) < len(keys)
        fan_in = None
        if shared:
            assert len(set(keys)) == 1, "BatchNorm launch mixes shared and private inputs"
            x = self.segs[0]["x"]
            fan_in = plan.new(x.cols * len(self.segs), rows=x.rows)
        ragged_acc = {}
        for c0 in range(0, len(self.segs), L.MAX_BN_SEGS):
            chunk = self.segs[c0:c0 + L.MAX_BN_SEGS]
            a = L.BnBwdArgs()
            a.n_seg, a.training, a.relu = len(chunk), 1 if plan.training else 0, 1 if self.relu else 0
            a.eps = self.eps
            a.mask_scale = plan.keep_scale if self.drop_p > 0 else 1.0
            a.M = self.M
            a.row_offsets = None if self.row_offsets is None else self.row_offsets.data_ptr()
            a.workspace = plan._bn_ws.data_ptr()
            pre = []
            for i, s in enumerate(chunk):
                S = a.s[i]
                plan.ensure_grad(s["y"], gs)
                yg = s["y"].grad
                S.dy, S.lddy = yg.ptr, yg.ld
                S.half = 0
                if plan.is_half_only(s["y"]):
                    S.y, S.ldy = plan.shadow_view(s["y"])
                    S.half |= L.BN_Y_BF16
                else:
                    S.y, S.ldy = s["y"].ptr, s["y"].ld
                S.x, S.ldx = s["x"].ptr, s["x"].ld
                S.accumulate_dx = 0
                S.dxh = None
                if fan_in is not None:
                    sl = fan_in.slice((c0 + i) * s["x"].cols, (c0 + i + 1) * s["x"].cols)
                    S.dx, S.lddx = sl.ptr, sl.ld
                else:
                    xg = s["x"].grad
                    key = keys[c0 + i]
                    if self.row_offsets is not None:
                        if key not in ragged_acc:
                            ragged_acc[key] = gs.claim(s["x"])
                        S.accumulate_dx = 1 if ragged_acc[key] else 0
                    elif gs.claim(s["x"]):
                        S.accumulate_dx = 1
                    S.dx, S.lddx = xg.ptr, xg.ld
                    if plan.use_g2 and plan.shadow_wanted(xg):
                        S.dxh, S.lddxh = plan.shadow_view(xg)
                        plan.mark_shadow(xg)
                        # x is the output of a bf16 linear launch whose backward reads dZ through the shadow only
                        if s.get("_lin_g2") and not S.accumulate_dx and self.row_offsets is None:
                            plan.make_grad_half_only(s["x"])
                            S.dx = None
                S.gamma = s["gamma"].data_ptr()
                if plan.training:
                    S.save_mean, S.save_invstd = s["save_mean"].data_ptr(), s["save_invstd"].data_ptr()
                else:
                    inv = s["save_invstd"]
                    rm, rv = s["running_mean"], s["running_var"]
                    pre.append(lambda st, inv=inv, rv=rv, eps=self.eps: inv.copy_(torch.rsqrt(rv + eps)))
                    S.save_mean, S.save_invstd = rm.data_ptr(), inv.data_ptr()
                if s.get("gamma_param") is not None:
                    gg, gb = plan.param_grad(s["gamma_param"]), plan.param_grad(s["beta_param"])
                    if plan._claim_param(s["gamma_param"]) or plan._claim_param(s["beta_param"]):
                        raise RuntimeError("BatchNorm parameters used twice in one plan")
                    S.dgamma, S.dbeta = gg.data_ptr(), gb.data_ptr()
                else:
                    S.dgamma = None if s.get("dgamma") is None else s["dgamma"].data_ptr()
                    S.dbeta = None if s.get("dbeta") is None else s["dbeta"].data_ptr()
                S.C = s["x"].cols
                S.row_group = s.get("row_group", 0)
            self._keep.append(a)
            plan.bwd_steps.extend(pre)
            if plan.dist is not None and plan.training:
                total_c = sum(s["x"].cols for s in chunk)
                ex = torch.zeros(2 * total_c + len(chunk), dtype=torch.float64, device=plan.device)
                a.phase, a.exchange = 1, ex.data_ptr()
                a2 = L.BnBwdArgs.from_buffer_copy(a)
                a2.phase = 2
                self._keep += [ex, a2]
                plan.bwd_steps.append(plan.call("cdc_bn_bwd", C.byref(a), what="cdc_bn_bwd(stats)"))
                plan.bwd_steps.append(plan.comm(lambda ex=ex: plan.dist.all_reduce_sum(ex)))
                plan.bwd_steps.append(plan.call("cdc_bn_bwd", C.byref(a2), what="cdc_bn_bwd(apply)"))
            else:
                plan.bwd_steps.append(plan.call("cdc_bn_bwd", C.byref(a)))
        if fan_in is not None:
            x = self.segs[0]["x"]
            if x.mask is not None:
                raise RuntimeError("shared BatchNorm input cannot be an activation-fused linear output")
            acc = 1 if gs.claim(x) else 0
            xg = x.grad
            plan.bwd_steps.append(plan.call("cdc_sum_slices", fan_in.cptr(), C.c_int64(fan_in.ld), xg.cptr(), C.c_int64(xg.ld),
                                            C.c_int64(x.rows), x.cols, len(self.segs), acc))


class RowDot:
    """out[b] = sigmoid?(x[b,:]·w + bias + sum addends[b])  for several groups (towers) in one launch.

    groups: list of dicts {x: Buf, w: Parameter [1,K] or [K], b: Parameter [1] or None, out: Buf [B,1]}
    addends: list of Buf [B,1] added to every group's logit (model/layer.py:53-54 `y_logits += other`).
    """

    def __init__(self, plan, groups, addends=(), sigmoid=False, row_offsets=None, M=None):
        self.groups, self.addends, self.sigmoid = groups, list(addends), sigmoid
        self.row_offsets = row_offsets
        self.M = plan.B if M is None else M
        assert len(groups) <= L.MAX_GROUPS and len(self.addends) <= 4
        kmax = 0
        for g in groups:
            K = g["w"].numel()
            assert g["x"].cols == K
            if g.get("out") is None:
                g["out"] = plan.new(1, rows=g["x"].rows)
            kmax = max(kmax, K)
        self.kmax = kmax
        self.outs = [g["out"] for g in groups]
        plan.need_rowdot_ws(len(groups) * L.ROWDOT_PARTS * (kmax + 1))
        if self.addends:
            if row_offsets is not None:
                shared = plan.new(1)                      # ragged groups own disjoint rows of one buffer
                self.dlogit = [shared for _ in groups]
            else:
                self.dlogit = [plan.new(1) for _ in groups]
        self._keep = []
        plan.add(self)

    def build_fwd(self, plan):
        a = L.RowdotFwdArgs()
        a.n_groups, a.sigmoid, a.n_addend, a.M = len(self.groups), 1 if self.sigmoid else 0, len(self.addends), self.M
        a.row_offsets = None if self.row_offsets is None else self.row_offsets.data_ptr()
        for i, ad in enumerate(self.addends):
            a.addend[i], a.ld_addend[i] = ad.ptr, ad.ld
        for i, g in enumerate(self.groups):
            G = a.g[i]
            G.x, G.ldx = g["x"].ptr, g["x"].ld
            G.w = g["w"].data_ptr()
            G.bias = None if g.get("b") is None else g["b"].data_ptr()
            G.out, G.ld_out = g["out"].ptr, g["out"].ld
            G.logit = None
            G.K = g["w"].numel()
        self._keep.append(a)
        plan.fwd_steps.append(plan.call("cdc_rowdot_fwd", C.byref(a)))

    def build_bwd(self, plan, gs):
        # groups that read the SAME x (CrossNetMix's gate scores) would race on dx inside one launch: they go out
        # in successive launches; groups with private inputs (towers) share one
        waves = []
        for i, g in enumerate(self.groups):
            key = (id(g["x"].root), g["x"].col0, g["x"].cols)
            for w in waves:
                if self.row_offsets is not None or key not in w["keys"]:
                    w["keys"].add(key)
                    w["idx"].append(i)
                    break
            else:
                waves.append({"keys": {key}, "idx": [i]})
        ragged_acc = {}
        for w in waves:
            self._build_bwd_launch(plan, gs, w["idx"], ragged_acc)

    def _build_bwd_launch(self, plan, gs, idxs, ragged_acc):
        a = L.RowdotBwdArgs()
        a.n_groups, a.sigmoid, a.M = len(idxs), 1 if self.sigmoid else 0, self.M
        first = idxs[0]
        a.row_offsets = None if self.row_offsets is None else self.row_offsets.data_ptr() + 4 * first
        if self.row_offsets is not None:
            assert idxs == list(range(first, first + len(idxs)))
        a.workspace = plan._rowdot_ws.data_ptr()
        post = []
        for slot, i in enumerate(idxs):
            g = self.groups[i]
            G = a.g[slot]
            plan.ensure_grad(g["out"], gs)
            og = g["out"].grad
            G.dout, G.ld_dout = og.ptr, og.ld
            G.out, G.ld_out = g["out"].ptr, g["out"].ld
            G.x, G.ldx = g["x"].ptr, g["x"].ld
            G.w = g["w"].data_ptr()
            if g.get("no_dx"):
                G.dx = None
            else:
                if g["x"].mask is not None:
                    raise RuntimeError("rowdot cannot consume an activation-fused linear output")
                xg = g["x"].grad
                G.dx, G.lddx = xg.ptr, xg.ld
                if self.row_offsets is not None:
                    key = (id(g["x"].root), g["x"].col0, g["x"].cols)
                    if key not in ragged_acc:
                        ragged_acc[key] = gs.claim(g["x"])
                    G.accumulate_dx = 1 if ragged_acc[key] else 0
                else:
                    G.accumulate_dx = 1 if gs.claim(g["x"]) else 0
            gw = plan.param_grad(g["w"])
            if plan._claim_param(g["w"]):
                # the same weight is used by an earlier-built op (CrossNetMix gates are shared by every layer):
                # write this use's gradient to a temporary and add it afterwards
                tmp = torch.zeros_like(gw)
                self._keep.append(tmp)
                G.dw = tmp.data_ptr()
                post.append(plan.call("cdc_copy_or_add", _p(gw), C.c_int64(gw.numel()), _p(tmp), C.c_int64(gw.numel()),
                                      C.c_int64(1), gw.numel(), 1))
            else:
                G.dw = gw.data_ptr()
            if g.get("b") is not None:
                gb = plan.param_grad(g["b"])
                plan._claim_param(g["b"])
                G.dbias = gb.data_ptr()
            else:
                G.dbias = None
            if self.addends:
                G.dlogit, G.ld_dlogit = self.dlogit[i].ptr, self.dlogit[i].ld
                if self.row_offsets is not None and i == 0:
                    for ad in self.addends:
                        adg = ad.grad
                        if gs.claim(ad):
                            post.append(plan.call("cdc_add_inplace", adg.cptr(), C.c_int64(adg.ld), self.dlogit[i].cptr(),
                                                  C.c_int64(self.dlogit[i].ld), C.c_int64(self.M), 1))
                        else:
                            t_dst, t_src = adg.tensor(), self.dlogit[i].tensor()
                            post.append(lambda st, d=t_dst, s_=t_src: d.copy_(s_))
            else:
                G.dlogit = None
            G.K = g["w"].numel()
        if self.addends and self.row_offsets is None:
            # every group's logit gradient flows into every addend: one fan-in launch per addend, this launch's groups added in order
            for ad in self.addends:
                adg = ad.grad
                n = L.AddNArgs()
                n.dst, n.ld_dst, n.rows, n.cols, n.n = adg.ptr, adg.ld, self.M, 1, len(idxs)
                n.accumulate = 1 if gs.claim(ad) else 0
                for k, i in enumerate(idxs):
                    n.src[k], n.ld_src[k] = self.dlogit[i].ptr, self.dlogit[i].ld
                self._keep.append(n)
                post.append(plan.call("cdc_add_n", C.byref(n)))
        self._keep.append(a)
        self.bwd_args = getattr(self, "bwd_args", []) + [a]       # (trainer: the fused BCE is switched on in the head's launch)
        plan.bwd_steps.append(plan.call("cdc_rowdot_bwd", C.byref(a)))
        plan.bwd_steps.extend(post)


class TowerHead:
    """The tower head in one launch per direction (csrc/head.hip; model/layer.py:48-56): per tower the output Linear(->1), plus the
    wide term formed once per row, plus further [B,1] logits, sigmoid, column t of `out`.  The backward also forms the wide
    term's gradient (the sum of the towers' logit gradients) and, with the trainer's fused loss, BCELoss and its gradient.

    towers: list of dicts {x: Buf [B,K], w: Parameter [1,K] or [K], b: Parameter [1] or None};  out: Buf [B, n_tower];
    wide: dict {x: Buf [B,Kw], w, b} or None;  addends: Bufs [B,1]."""

    def __init__(self, plan, towers, out, wide=None, addends=(), sigmoid=True):
        assert 0 < len(towers) <= L.HEAD_MAX_TOWERS and len(addends) <= 2 and out.cols == len(towers)
        self.towers, self.out, self.wide, self.addends, self.sigmoid = towers, out, wide, list(addends), sigmoid
        self.M = plan.B
        for t in towers:
            assert t["x"].cols == t["w"].numel()
            if t["x"].mask is not None:
                raise RuntimeError("the tower head cannot consume an activation-fused linear output")
        if wide is not None:
            assert wide["x"].cols == wide["w"].numel()
        width = sum(t["w"].numel() + 1 for t in towers) + (wide["w"].numel() + 1 if wide is not None else 0)
        self.ws = torch.empty(L.ROWDOT_PARTS * width, dtype=torch.float32, device=plan.device)
        self._keep = []
        plan.add(self)

    def _fill(self, a, plan):
        a.n_tower, a.sigmoid, a.n_addend, a.M = len(self.towers), 1 if self.sigmoid else 0, len(self.addends), self.M
        a.out, a.ld_out = self.out.ptr, self.out.ld
        for i, t in enumerate(self.towers):
            T = a.t[i]
            T.x, T.ldx = t["x"].ptr, t["x"].ld
            T.w = t["w"].data_ptr()
            T.bias = None if t.get("b") is None else t["b"].data_ptr()
            T.K = t["w"].numel()
        if self.wide is not None:
            w = self.wide
            a.wide_x, a.ld_wide = w["x"].ptr, w["x"].ld
            a.wide_w = w["w"].data_ptr()
            a.wide_bias = None if w.get("b") is None else w["b"].data_ptr()
            a.wide_K = w["w"].numel()
        for i, ad in enumerate(self.addends):
            a.addend[i], a.ld_addend[i] = ad.ptr, ad.ld

    def build_fwd(self, plan):
        a = L.HeadArgs()
        self._fill(a, plan)
        self._keep.append(a)
        step = plan.call("cdc_head_fwd", C.byref(a))
        self._fwd_calls = [step]
        plan.fwd_steps.append(step)

    def build_bwd(self, plan, gs):
        a = L.HeadArgs()
        self._fill(a, plan)
        plan.ensure_grad(self.out, gs)
        og = self.out.grad
        a.d_out, a.ld_dout = og.ptr, og.ld
        for i, t in enumerate(self.towers):
            T = a.t[i]
            xg = t["x"].grad
            T.accumulate_dx = 1 if gs.claim(t["x"]) else 0
            T.dx, T.lddx = xg.ptr, xg.ld
            if plan._claim_param(t["w"]) or (t.get("b") is not None and plan._claim_param(t["b"])):
                raise RuntimeError("a tower's output layer is used twice in one plan")
            T.dw = plan.param_grad(t["w"]).data_ptr()
            T.dbias = None if t.get("b") is None else plan.param_grad(t["b"]).data_ptr()
        if self.wide is not None:
            w = self.wide
            if w["x"].mask is not None:
                raise RuntimeError("the wide term cannot consume an activation-fused linear output")
            xg = w["x"].grad
            a.accumulate_wide_dx = 1 if gs.claim(w["x"]) else 0
            a.wide_dx, a.ld_wide_dx = xg.ptr, xg.ld
            if plan._claim_param(w["w"]) or (w.get("b") is not None and plan._claim_param(w["b"])):
                raise RuntimeError("the wide term's parameters are used twice in one plan")
            a.wide_dw = plan.param_grad(w["w"]).data_ptr()
            a.wide_dbias = None if w.get("b") is None else plan.param_grad(w["b"]).data_ptr()
        for i, ad in enumerate(self.addends):
            adg = ad.grad
            a.accumulate_d_addend[i] = 1 if gs.claim(ad) else 0
            a.d_addend[i], a.ld_d_addend[i] = adg.ptr, adg.ld
        a.workspace = self.ws.data_ptr()
        self._keep.append(a)
        self.bwd_args = [a]                      # (trainer: the fused BCE is switched on in this launch)
        plan.bwd_steps.append(plan.call("cdc_head_bwd", C.byref(a)))


class TowerChain:
    """The towers of a multi-tower model as ONE launch per direction (csrc/tower.hip; model/layer.py:35-56,178-206): built from
    the five ops the model described — GLinear, BatchNorm, GLinear, BatchNorm, TowerHead — where they have the instantiated
    shape; it takes their place in plan.ops and uses their buffers, seeds and parameters.  The grad-weight contractions of the two
    linear layers stay with the batched launch at the end of backward (the backward launch writes their dZ operands as bf16)."""

    SHAPES = ((64, 64, 32), (128, 64, 32))      # (H0, H1, H2) csrc/tower.hip is instantiated for (config.py:39-42: tower_dims (64, 32))
    enabled = True                              # tests set this to False to build the five launches at a matching shape
    # set by the trainer around a step whose loss is formed inside the backward launch (fused BCE, one GPU): the forward step then
    # launches nothing and the backward step launches cdc_tower_step — forward and backward of the towers in ONE launch.  The
    # tower forward is the last launch of the plan's forward and the tower backward the first of its backward, so nothing moves.
    one_launch = False

    @classmethod
    def match(cls, plan, l1, b1, l2, b2, head):
        if not (isinstance(l1, GLinear) and isinstance(b1, BatchNorm) and isinstance(l2, GLinear) and isinstance(b2, BatchNorm) and
                isinstance(head, TowerHead)):
            return False
        # data parallel with global-batch statistics (plan.dist): the split form, cdc_tower_dp — six phases with the BatchNorm sums
        # all-reduced between them; a rank's share may be a single row there (the GLOBAL batch decides about BatchNorm)
        if not (plan.use_g2 and plan.training and (plan.B >= 2 or plan.dist is not None)):
            return False
        n = len(head.towers)
        if not (0 < n <= L.TOWER_MAX and n * math.ceil(plan.B / L.TOWER_ROWS) <= 256):
            return False
        if head.addends or not head.sigmoid or head.M != plan.B:
            return False
        for lin in (l1, l2):
            if not (lin.g2 and lin.row_offsets is None and lin.M == plan.B and not lin.adopted and not lin.relu and lin.drop_p == 0.0 and
                    len(lin.groups) == n and all(isinstance(g["w"], torch.Tensor) and g.get("b") is not None and g["act_cols"] == 0
                                                  for g in lin.groups)):
                return False
        for bn in (b1, b2):
            if not (len(bn.segs) == n and bn.row_offsets is None and bn.M == plan.B and bn.relu and not bn.skip_le1 and
                    all(s.get("gamma_param") is not None and s.get("row_group", 0) == 0 for s in bn.segs)):
                return False
        if b1.drop_p != b2.drop_p or b1.eps != b2.eps or b1.momentum != b2.momentum:
            return False
        H1, H0 = l1.groups[0]["w"].shape
        H2 = l2.groups[0]["w"].shape[0]
        if (H0, H1, H2) not in cls.SHAPES:
            return False
        same = CGCMid._same
        for i in range(n):
            g1, g2, s1, s2, t = l1.groups[i], l2.groups[i], b1.segs[i], b2.segs[i], head.towers[i]
            if not (tuple(g1["w"].shape) == (H1, H0) and tuple(g2["w"].shape) == (H2, H1) and t["w"].numel() == H2):
                return False
            if not (same(g1["y"], s1["x"]) and same(s1["y"], g2["x"]) and same(g2["y"], s2["x"]) and same(s2["y"], t["x"])):
                return False
        if head.wide is not None:
            w = head.wide
            K = w["w"].numel()
            if not (K <= 512 and K % 4 == 0 and w["x"].ld % 4 == 0 and w["x"].ptr % 16 == 0 and w["x"].cols == K):
                return False
        return True

    def __init__(self, plan, l1, b1, l2, b2, head):
        self.l1, self.b1, self.l2, self.b2, self.head = l1, b1, l2, b2, head
        i = plan.ops.index(l1)
        assert plan.ops[i:i + 5] == [l1, b1, l2, b2, head]
        plan.ops[i:i + 5] = [self]
        self.sigmoid, self.out, self.M = head.sigmoid, head.out, head.M
        self.n = len(head.towers)
        self.H1, self.H0 = l1.groups[0]["w"].shape
        self.H2 = l2.groups[0]["w"].shape[0]
        self.tmo_word = torch.zeros(1, dtype=torch.int32, device=plan.device)     # CDC_TOWER_ERR_TIMEOUT lands here (TrainStep.check_ids)
        self.ws = None
        self._keep = []
        for op in (l1, l2):
            op._keep = getattr(op, "_keep", [])
        # the hidden activation is read by the second contraction and the grad-weight launch only: bf16 alone
        for s in b1.segs:
            plan.want_shadow(s["y"])
            plan.make_value_half_only(s["y"])

    def _fill(self, a, plan):
        head, b1 = self.head, self.b1
        a.n_tower, a.H0, a.H1, a.H2, a.M = self.n, self.H0, self.H1, self.H2, plan.B
        a.relu, a.sigmoid = 1, 1 if head.sigmoid else 0
        a.drop_p, a.eps, a.momentum = b1.drop_p, b1.eps, b1.momentum
        a.seed1, a.seed2 = self.b1.seed & 0xFFFFFFFFFFFFFFFF, self.b2.seed & 0xFFFFFFFFFFFFFFFF
        a.seed_offset_dev = plan.step_dev.data_ptr()
        a.out, a.ld_out = head.out.ptr, head.out.ld
        if head.wide is not None:
            w = head.wide
            a.wide_x, a.ld_wide = w["x"].ptr, w["x"].ld
            a.wide_w = w["w"].data_ptr()
            a.wide_bias = None if w.get("b") is None else w["b"].data_ptr()
            a.wide_K = w["w"].numel()
        a.err = self.tmo_word.data_ptr()
        for i in range(self.n):
            T = a.t[i]
            g1, g2, s1, s2, t = self.l1.groups[i], self.l2.groups[i], self.b1.segs[i], self.b2.segs[i], head.towers[i]
            T.xh, T.ldxh = plan.shadow_view(g1["x"])
            for Y, g, s in ((T.l1, g1, s1), (T.l2, g2, s2)):
                wh, wt = plan.wshadow(g["w"])
                Y.wh, Y.ldwh = wh.data_ptr(), wh.stride(0)
                Y.wt, Y.ldwt = wt.data_ptr(), wt.stride(0)
                Y.bias = g["b"].data_ptr()
                Y.z, Y.ldz = g["y"].ptr, g["y"].ld
                Y.gamma, Y.beta = s["gamma"].data_ptr(), s["beta"].data_ptr()
                Y.running_mean, Y.running_var = s["running_mean"].data_ptr(), s["running_var"].data_ptr()
                nbt = s.get("num_batches_tracked")
                Y.num_batches_tracked = None if nbt is None else nbt.data_ptr()
                Y.save_mean, Y.save_invstd = s["save_mean"].data_ptr(), s["save_invstd"].data_ptr()
            T.a1h, T.lda1h = plan.shadow_view(s1["y"])
            T.a2, T.lda2 = s2["y"].ptr, s2["y"].ld
            T.wo = t["w"].data_ptr()
            T.bo = None if t.get("b") is None else t["b"].data_ptr()
        if self.ws is None:
            need = int(self.lib_ws_bytes(plan, a))
            self.ws = torch.zeros(need, dtype=torch.uint8, device=plan.device)
        a.workspace = self.ws.data_ptr()

    @staticmethod
    def lib_ws_bytes(plan, a):
        n = plan.lib.cdc_tower_workspace_bytes(C.byref(a))
        if n <= 0:
            raise RuntimeError("cdc_tower_workspace_bytes refused the argument block")
        return n

    def _dp_buffers(self, a, plan):
        """the split form's exchange buffers ([2 n C column sums | n row counts] doubles per exchange) and gradient scratch"""
        if getattr(self, "_ex", None) is None:
            n = self.n
            self._ex = [torch.zeros(2 * n * c + n, dtype=torch.float64, device=plan.device) for c in (self.H1, self.H2, self.H2, self.H1)]
            self._dy2, self._dy1 = plan.new(n * self.H2), plan.new(n * self.H1)
        for e in range(4):
            a.exchange[e] = self._ex[e].data_ptr()
        for i in range(self.n):
            T = a.t[i]
            y2, y1 = self._dy2.slice(i * self.H2, (i + 1) * self.H2), self._dy1.slice(i * self.H1, (i + 1) * self.H1)
            T.dy2, T.lddy2, T.dy1, T.lddy1 = y2.ptr, y2.ld, y1.ptr, y1.ld

    def _dp_steps(self, plan, a, phases, exchanges, steps, flops):
        for k, ph in enumerate(phases):
            steps.append(plan.call("cdc_tower_dp", C.byref(a), ph, what=f"cdc_tower_dp({ph})", flops=flops if k == 0 else 0.0))
            if k < len(exchanges):
                ex = self._ex[exchanges[k]]
                steps.append(plan.comm(lambda ex=ex: plan.dist.all_reduce_sum(ex)))

    def build_fwd(self, plan):
        plan.ensure_shadows([g["x"] for g in self.l1.groups], plan.fwd_steps)
        a = L.TowerArgs()
        self._fill(a, plan)
        for s in self.b1.segs:
            plan.mark_shadow(s["y"])
        self._keep.append(a)
        fl = sum(2.0 * plan.B * g["w"].shape[0] * g["w"].shape[1] for g in self.l1.groups + self.l2.groups)
        if plan.dist is not None:
            # global-batch BatchNorm statistics: phases 1 | all-reduce | 2 | all-reduce | 3 (csrc/tower.hip cdc_tower_dp)
            self._dp_buffers(a, plan)
            self._dp_steps(plan, a, (1, 2, 3), (0, 1), plan.fwd_steps, fl)
            return
        fwd = plan.call("cdc_tower_fwd", C.byref(a), flops=fl)

        def step(stream):
            if not self.one_launch:               # (one_launch: both directions go out where the backward stands, cdc_tower_step)
                fwd(stream)
        self._fwd_calls = [step]
        plan.fwd_steps.append(step)

    def build_bwd(self, plan, gs):
        head = self.head
        a = L.TowerArgs()
        self._fill(a, plan)
        plan.ensure_grad(head.out, gs)
        og = head.out.grad
        a.d_out, a.ld_dout = og.ptr, og.ld
        for i in range(self.n):
            T = a.t[i]
            g1, g2, s1, s2, t = self.l1.groups[i], self.l2.groups[i], self.b1.segs[i], self.b2.segs[i], head.towers[i]
            if plan._claim_param(t["w"]) or (t.get("b") is not None and plan._claim_param(t["b"])):
                raise RuntimeError("a tower's output layer is used twice in one plan")
            T.dwo = plan.param_grad(t["w"]).data_ptr()
            T.dbo = None if t.get("b") is None else plan.param_grad(t["b"]).data_ptr()
            for Y, g, s in ((T.l2, g2, s2), (T.l1, g1, s1)):
                if plan._claim_param(s["gamma_param"]) or plan._claim_param(s["beta_param"]):
                    raise RuntimeError("BatchNorm parameters used twice in one plan")
                Y.dgamma, Y.dbeta = plan.param_grad(s["gamma_param"]).data_ptr(), plan.param_grad(s["beta_param"]).data_ptr()
                # dZ of the layer: bf16 alone, read by the batched grad-weight launch
                if gs.claim(g["y"]):
                    raise RuntimeError("TowerChain: a tower layer's output has another gradient writer")
                plan.make_grad_half_only(g["y"])
                Y.dzh, Y.lddzh = plan.shadow_view(g["y"].grad)
                plan.mark_shadow(g["y"].grad)
            x = g1["x"]
            T.accumulate_dx = 1 if gs.claim(x) else 0
            xg = x.grad
            if plan.is_half_only(xg):
                raise RuntimeError("TowerChain: the tower input's gradient is kept as bf16 only")
            T.dx, T.lddx = xg.ptr, xg.ld
        if head.wide is not None:
            w = head.wide
            if w["x"].mask is not None:
                raise RuntimeError("the wide term cannot consume an activation-fused linear output")
            xg = w["x"].grad
            a.accumulate_wide_dx = 1 if gs.claim(w["x"]) else 0
            a.wide_dx, a.ld_wide_dx = xg.ptr, xg.ld
            if plan._claim_param(w["w"]) or (w.get("b") is not None and plan._claim_param(w["b"])):
                raise RuntimeError("the wide term's parameters are used twice in one plan")
            a.wide_dw = plan.param_grad(w["w"]).data_ptr()
            a.wide_dbias = None if w.get("b") is None else plan.param_grad(w["b"]).data_ptr()
        self._keep.append(a)
        self.bwd_args = [a]                      # (trainer: the fused BCE is switched on in this launch)
        fl = sum(2.0 * plan.B * g["w"].shape[0] * g["w"].shape[1] for g in self.l1.groups + self.l2.groups)
        if plan.dist is not None:
            self._dp_buffers(a, plan)
            self._dp_steps(plan, a, (4, 5, 6), (2, 3), plan.bwd_steps, fl)
        else:
            bwd = plan.call("cdc_tower_bwd", C.byref(a), flops=fl)
            both = plan.call("cdc_tower_step", C.byref(a), flops=2.0 * fl)

            def step(stream):
                (both if self.one_launch else bwd)(stream)
            plan.bwd_steps.append(step)
        # the grad-weight contractions of both layers: batched launch at the end of backward (operands: the shadows written above)
        for lin in (self.l2, self.l1):
            lin.skip_bwd_x = True
            lin.build_bwd(plan, gs)


def fuse_tower(plan, enable=True):
    """Replaces the run (GLinear, BatchNorm, GLinear, BatchNorm, TowerHead) of plan.ops that is the towers of a multi-tower model in
    the instantiated shape by ONE TowerChain op (training plans on one GPU; every other shape, precision or mode keeps the five
    ops)."""
    if not (enable and TowerChain.enabled):
        return 0
    n = 0
    i = 0
    while i + 4 < len(plan.ops):
        if TowerChain.match(plan, *plan.ops[i:i + 5]):
            TowerChain(plan, *plan.ops[i:i + 5])
            n += 1
        i += 1
    return n


class CrossLayer:
    """DCN-v1 cross layer (model/layer.py:321-329): out = x0 * (xl·w) + b + xl."""

    def __init__(self, plan, x0, xl, w, b, out=None):
        self.x0, self.xl, self.w, self.b = x0, xl, w, b
        self.out = out if out is not None else plan.new(x0.cols)
        self.xw = torch.empty(plan.B, dtype=torch.float32, device=plan.device)
        self.ws = torch.empty(L.ROWDOT_PARTS * 2 * x0.cols, dtype=torch.float32, device=plan.device)
        plan.add(self)

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_cross_fwd", self.x0.cptr(), C.c_int64(self.x0.ld), self.xl.cptr(),
                                        C.c_int64(self.xl.ld), _p(self.w.data), _p(self.b.data), self.out.cptr(),
                                        C.c_int64(self.out.ld), _p(self.xw), C.c_int64(plan.B), self.x0.cols))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        og = self.out.grad
        if self.x0.mask is not None or self.xl.mask is not None:
            raise RuntimeError("a cross layer cannot consume an activation-fused linear output")
        # d_x0 accumulates inside the kernel: its gradient region must exist first
        plan.ensure_grad(self.x0, gs)
        x0g = self.x0.grad
        same = (self.xl.root is self.x0.root and self.xl.col0 == self.x0.col0)
        if same:
            dxl = plan.new(self.x0.cols)
        else:
            if gs.claim(self.xl):
                raise RuntimeError("a cross layer input (x_l) feeds more than one consumer")
            dxl = self.xl.grad
        gw, gb = plan.param_grad(self.w), plan.param_grad(self.b)
        plan._claim_param(self.w), plan._claim_param(self.b)
        plan.bwd_steps.append(plan.call("cdc_cross_bwd", og.cptr(), C.c_int64(og.ld), self.x0.cptr(), C.c_int64(self.x0.ld),
                                        self.xl.cptr(), C.c_int64(self.xl.ld), _p(self.w.data), _p(self.xw), x0g.cptr(),
                                        C.c_int64(x0g.ld), dxl.cptr(), C.c_int64(dxl.ld), _p(gw), _p(gb), _p(self.ws),
                                        C.c_int64(plan.B), self.x0.cols))
        if same:
            plan.bwd_steps.append(plan.call("cdc_add_inplace", x0g.cptr(), C.c_int64(x0g.ld), dxl.cptr(), C.c_int64(dxl.ld),
                                            C.c_int64(plan.B), self.x0.cols))


class Tanh:
    """y = tanh(x) over a whole buffer (CrossNetMix, model/layer.py:387,389)."""

    def __init__(self, plan, x, out=None):
        self.x = x
        self.out = out if out is not None else plan.new(x.cols, rows=x.rows)
        plan.add(self)

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_tanh_fwd", self.x.cptr(), C.c_int64(self.x.ld), self.out.cptr(), C.c_int64(self.out.ld),
                                        C.c_int64(self.x.rows), self.x.cols))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        if self.x.mask is not None:
            raise RuntimeError("tanh cannot consume an activation-fused linear output")
        acc = gs.claim(self.x)
        og, xg = self.out.grad, self.x.grad
        plan.bwd_steps.append(plan.call("cdc_tanh_bwd", og.cptr(), C.c_int64(og.ld), self.out.cptr(), C.c_int64(self.out.ld), xg.cptr(),
                                        C.c_int64(xg.ld), C.c_int64(self.x.rows), self.x.cols, 1 if acc else 0))


class Reshape:
    """The same values under another [rows, cols] geometry (a copy: e.g. [B, F*D] embeddings as [B*F, D] field tokens,
    model/layer.py:72).  The source must be a whole contiguous buffer."""

    def __init__(self, plan, x, rows, cols, out=None):
        """out: optional destination [rows, cols] with any row stride (e.g. a column slice of a concatenation buffer)"""
        assert x.col0 == 0 and x.ld == x.cols and x.rows * x.cols == rows * cols, "reshape needs a whole contiguous buffer"
        self.x, self.rows, self.cols = x, rows, cols
        self.out = out if out is not None else plan.new(cols, rows=rows)
        assert self.out.rows == rows and self.out.cols == cols
        if x.mask is not None:
            raise RuntimeError("reshape cannot consume an activation-fused linear output")
        plan.add(self)

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_copy_or_add", self.out.cptr(), C.c_int64(self.out.ld), self.x.cptr(), C.c_int64(self.cols),
                                        C.c_int64(self.rows), self.cols, 0))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        acc = gs.claim(self.x)
        og, xg = self.out.grad, self.x.grad
        plan.bwd_steps.append(plan.call("cdc_copy_or_add", xg.cptr(), C.c_int64(self.cols), og.cptr(), C.c_int64(og.ld),
                                        C.c_int64(self.rows), self.cols, 1 if acc else 0))


class AttnCore:
    """Per-sample multi-head self-attention over F tokens (the inside of nn.MultiheadAttention, model/layer.py:75-77):
    qkv [B*F, 3A] -> [B*F, A]; dropout on the probabilities in training."""

    def __init__(self, plan, qkv, n_tokens, n_head):
        assert qkv.cols % 3 == 0 and qkv.rows % n_tokens == 0
        self.qkv, self.F, self.H = qkv, n_tokens, n_head
        self.A = qkv.cols // 3
        self.Bs = qkv.rows // n_tokens
        self.out = plan.new(self.A, rows=qkv.rows)
        self.probs = torch.empty((self.Bs, n_head, n_tokens, n_tokens), dtype=torch.float32, device=plan.device)
        self.drop_p = plan.dropout if plan.training else 0.0
        self.seed = plan._next_seed()
        if qkv.mask is not None:
            raise RuntimeError("attention cannot consume an activation-fused linear output")
        plan.add(self)

    def _tail(self, plan):
        return (C.c_int64(self.Bs), self.F, self.A, self.H, C.c_float(self.drop_p), C.c_uint64(self.seed), _p(plan.step_dev))

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_attn_fwd", self.qkv.cptr(), C.c_int64(self.qkv.ld), self.out.cptr(), C.c_int64(self.out.ld),
                                        _p(self.probs), *self._tail(plan)))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        if gs.claim(self.qkv):
            raise RuntimeError("the attention core must be the only consumer of its qkv buffer")
        og, qg = self.out.grad, self.qkv.grad
        plan.bwd_steps.append(plan.call("cdc_attn_bwd", self.qkv.cptr(), C.c_int64(self.qkv.ld), _p(self.probs), og.cptr(),
                                        C.c_int64(og.ld), qg.cptr(), C.c_int64(qg.ld), *self._tail(plan)))


class AddRelu:
    """out = relu(a + b) (model/layer.py:80-82)."""

    def __init__(self, plan, a, b):
        assert a.rows == b.rows and a.cols == b.cols
        self.a, self.b = a, b
        self.out = plan.new(a.cols, rows=a.rows)
        if a.mask is not None or b.mask is not None:
            raise RuntimeError("add+relu cannot consume an activation-fused linear output")
        plan.add(self)

    def build_fwd(self, plan):
        a, b, o = self.a, self.b, self.out
        plan.fwd_steps.append(plan.call("cdc_add_relu_fwd", a.cptr(), C.c_int64(a.ld), b.cptr(), C.c_int64(b.ld), o.cptr(), C.c_int64(o.ld),
                                        C.c_int64(a.rows), a.cols))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        acc_a, acc_b = gs.claim(self.a), gs.claim(self.b)
        o, og, ag, bg = self.out, self.out.grad, self.a.grad, self.b.grad
        plan.bwd_steps.append(plan.call("cdc_add_relu_bwd", o.cptr(), C.c_int64(o.ld), og.cptr(), C.c_int64(og.ld), ag.cptr(),
                                        C.c_int64(ag.ld), 1 if acc_a else 0, bg.cptr(), C.c_int64(bg.ld), 1 if acc_b else 0,
                                        C.c_int64(o.rows), o.cols))


class HostStep:
    """A few torch ops inside the launch sequence (they run on the current stream and are captured with the graph):
    fwd and/or bwd are callables without arguments.  In the backward the step runs where reverse op order puts it."""

    def __init__(self, plan, fwd=None, bwd=None):
        self.fwd, self.bwd = fwd, bwd
        plan.add(self)

    def build_fwd(self, plan):
        if self.fwd is not None:
            plan.fwd_steps.append(lambda stream, fn=self.fwd: fn())

    def build_bwd(self, plan, gs):
        if self.bwd is not None:
            plan.bwd_steps.append(lambda stream, fn=self.bwd: fn())


class CopyCols:
    """dst <- src for two [rows, cols] views of any row stride (one piece of a torch.cat); detach=True: no gradient flows
    back (`.detach()` in the reference, e.g. model/pepnet.py:79, model/adasparse.py:96)."""

    def __init__(self, plan, src, dst, detach=False):
        assert src.rows == dst.rows and src.cols == dst.cols
        self.src, self.dst, self.detach = src, dst, detach
        if src.mask is not None:
            raise RuntimeError("a concatenation piece cannot be an activation-fused linear output")
        plan.add(self)

    def build_fwd(self, plan):
        s_, d_ = self.src, self.dst
        plan.fwd_steps.append(plan.call("cdc_copy_or_add", d_.cptr(), C.c_int64(d_.ld), s_.cptr(), C.c_int64(s_.ld), C.c_int64(s_.rows), s_.cols, 0))

    def build_bwd(self, plan, gs):
        if self.detach:
            return
        plan.ensure_grad(self.dst, gs)
        acc = gs.claim(self.src)
        dg, sg = self.dst.grad, self.src.grad
        plan.bwd_steps.append(plan.call("cdc_copy_or_add", sg.cptr(), C.c_int64(sg.ld), dg.cptr(), C.c_int64(dg.ld), C.c_int64(sg.rows), sg.cols,
                                        1 if acc else 0))


class SigmoidGate:
    """out = a * (beta * sigmoid(alpha * p), zeroed where <= eps)  (AdaSparse's pruner, PEPNet's GateNN product)."""

    def __init__(self, plan, a, p, beta=2.0, alpha=1.0, eps=-1.0, detach_a=False, out=None):
        assert a.rows == p.rows and a.cols == p.cols
        self.a, self.p, self.beta, self.alpha, self.eps, self.detach_a = a, p, float(beta), float(alpha), float(eps), detach_a
        self.out = out if out is not None else plan.new(a.cols, rows=a.rows)
        if a.mask is not None or p.mask is not None:
            raise RuntimeError("the sigmoid gate cannot consume an activation-fused linear output")
        plan.add(self)

    def _consts(self):
        return C.c_float(self.beta), C.c_float(self.alpha), C.c_float(self.eps)

    def build_fwd(self, plan):
        a, p, o = self.a, self.p, self.out
        plan.fwd_steps.append(plan.call("cdc_sigmoid_gate_fwd", a.cptr(), C.c_int64(a.ld), p.cptr(), C.c_int64(p.ld), o.cptr(), C.c_int64(o.ld),
                                        C.c_int64(a.rows), a.cols, *self._consts()))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        a, p, og = self.a, self.p, self.out.grad
        acc_p = gs.claim(p)
        pg = p.grad
        if self.detach_a:
            da, ldda, acc_a = None, C.c_int64(0), 0
        else:
            acc_a = 1 if gs.claim(a) else 0
            da, ldda = a.grad.cptr(), C.c_int64(a.grad.ld)
        plan.bwd_steps.append(plan.call("cdc_sigmoid_gate_bwd", a.cptr(), C.c_int64(a.ld), p.cptr(), C.c_int64(p.ld), og.cptr(), C.c_int64(og.ld),
                                        da, ldda, acc_a, pg.cptr(), C.c_int64(pg.ld), 1 if acc_p else 0, C.c_int64(a.rows), a.cols,
                                        *self._consts()))


class SelectByGroup:
    """out[b] = the feature block of row b's own group (model/hinet.py:71-74)."""

    def __init__(self, plan, feas, n_group, H, out=None):
        assert feas.cols == n_group * H
        self.feas, self.n_group, self.H = feas, n_group, H
        self.group = torch.zeros(plan.B, dtype=torch.int64, device=plan.device)
        self.out = out if out is not None else plan.new(H)
        if feas.mask is not None:
            raise RuntimeError("group selection cannot consume an activation-fused linear output")
        plan.add(self)

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_group_select_fwd", self.feas.cptr(), C.c_int64(self.feas.ld), _p(self.group), self.out.cptr(),
                                        C.c_int64(self.out.ld), C.c_int64(plan.B), self.n_group, self.H))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        acc = gs.claim(self.feas)
        og, fg = self.out.grad, self.feas.grad
        plan.bwd_steps.append(plan.call("cdc_group_select_bwd", og.cptr(), C.c_int64(og.ld), _p(self.group), fg.cptr(), C.c_int64(fg.ld),
                                        C.c_int64(plan.B), self.n_group, self.H, 1 if acc else 0))


class FMInteraction:
    """Second-order FM term over the gathered embeddings (model/layer.py:160-175): [B, F*D] -> [B, 1]."""

    def __init__(self, plan, x, n_fields, dim, out=None):
        assert x.cols == n_fields * dim
        self.x, self.F, self.D = x, n_fields, dim
        self.out = out if out is not None else plan.new(1, rows=x.rows)
        plan.add(self)

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_fm_fwd", self.x.cptr(), C.c_int64(self.x.ld), self.out.cptr(), C.c_int64(self.out.ld),
                                        C.c_int64(self.x.rows), self.F, self.D))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        if self.x.mask is not None:
            raise RuntimeError("the FM term cannot consume an activation-fused linear output")
        acc = gs.claim(self.x)
        og, xg = self.out.grad, self.x.grad
        plan.bwd_steps.append(plan.call("cdc_fm_bwd", self.x.cptr(), C.c_int64(self.x.ld), og.cptr(), C.c_int64(og.ld), xg.cptr(),
                                        C.c_int64(xg.ld), C.c_int64(self.x.rows), self.F, self.D, 1 if acc else 0))


class MatmulRight:
    """y_g = x_g @ M_g with M_g stored [K, N] row-major (CrossNetMix's V: model/layer.py:384 `v_list[i][k].t() @ x`).
    The three contractions reuse the grouped-linear kernels with the operand roles rotated."""

    def __init__(self, plan, groups):
        self.groups = groups
        self.M = plan.B
        assert len(groups) <= L.MAX_GROUPS
        for g in groups:
            K, N = g["m"].shape
            assert g["x"].cols == K
            if g.get("out") is None:
                g["out"] = plan.new(N)
        self.outs = [g["out"] for g in groups]
        self._keep = []
        plan.add(self)

    def build_fwd(self, plan):
        a = L.LinBwdxArgs()
        a.n_out = a.n_seg = len(self.groups)
        a.mask_scale = 1.0
        a.row_offsets = None
        for i, g in enumerate(self.groups):
            K, N = g["m"].shape
            O, S = a.o[i], a.s[i]
            O.dx, O.lddx = g["out"].ptr, g["out"].ld
            O.mask_y, O.mask_cols, O.accumulate = None, 0, 0
            O.M, O.K = self.M, N
            S.dz, S.lddz = g["x"].ptr, g["x"].ld
            S.w, S.ldw = g["m"].data_ptr(), N
            S.N, S.out = K, i
        self._keep.append(a)
        fl = sum(2.0 * self.M * g["m"].shape[0] * g["m"].shape[1] for g in self.groups)
        plan.fwd_steps.append(plan.call("cdc_glinear_bwd_x", C.byref(a), plan.prec, what="cdc_glinear_bwd_x(matmul_right fwd)", flops=fl))

    def build_bwd(self, plan, gs):
        for g in self.groups:
            plan.ensure_grad(g["out"], gs)
        # dM = x^T @ dY  (grad-weight kernel with dz := x, x := dY)
        a = L.LinBwdwArgs()
        a.n_groups, a.split_k, a.workspace, a.row_offsets = len(self.groups), 1, None, None
        for i, g in enumerate(self.groups):
            K, N = g["m"].shape
            G = a.g[i]
            og = g["out"].grad
            G.dz, G.lddz = g["x"].ptr, g["x"].ld
            G.x, G.ldx = og.ptr, og.ld
            gm = plan.param_grad(g["m"])
            G.dw, G.lddw = gm.data_ptr(), N
            G.db = None
            G.M, G.N, G.K = self.M, K, N
            G.accumulate = 1 if plan._claim_param(g["m"]) else 0
        self._keep.append(a)
        plan.bwd_steps.append(plan.call("cdc_glinear_bwd_w", C.byref(a), plan.prec, what="cdc_glinear_bwd_w(matmul_right)"))
        # dX = dY @ M^T  (forward kernel with w := M as [N=K_in, K=N_out]); it cannot accumulate, so go through a temporary
        f = L.LinFwdArgs()
        f.n_groups, f.relu, f.drop_p, f.seed, f.seed_offset_dev, f.row_offsets = len(self.groups), 0, 0.0, 0, None, None
        post = []
        for i, g in enumerate(self.groups):
            K, N = g["m"].shape
            if g["x"].mask is not None:
                raise RuntimeError("matmul_right cannot consume an activation-fused linear output")
            G = f.g[i]
            og = g["out"].grad
            xg = g["x"].grad
            acc = gs.claim(g["x"])
            if acc:
                tmp = plan.new(K)
                dst = tmp
                post.append(plan.call("cdc_copy_or_add", xg.cptr(), C.c_int64(xg.ld), tmp.cptr(), C.c_int64(tmp.ld), C.c_int64(self.M), K, 1))
            else:
                dst = xg
            G.x, G.ldx = og.ptr, og.ld
            G.w, G.ldw = g["m"].data_ptr(), N
            G.bias = None
            G.y, G.ldy = dst.ptr, dst.ld
            G.M, G.N, G.K = self.M, K, N
            G.act_cols = 0
        self._keep.append(f)
        plan.bwd_steps.append(plan.call("cdc_glinear_fwd", C.byref(f), plan.prec, what="cdc_glinear_fwd(matmul_right bwd)"))
        plan.bwd_steps.extend(post)


class CrossCombine:
    """out[:, k*P+e] = x0[:, e] * (u[:, k*P+e] + b1[e]) + b2[e] + r[:, k*P+e]
    CrossNetV2 (n_rep=1, b2=b_l, r=x_l: model/layer.py:342) and CrossNetMix's `x_0 * (uv_x + bias)` (:393-394)."""

    def __init__(self, plan, x0, u, b1=None, b2=None, r=None, n_rep=1, out=None):
        self.x0, self.u, self.b1, self.b2, self.r, self.n_rep = x0, u, b1, b2, r, n_rep
        self.P = x0.cols
        assert u.cols == self.P * n_rep
        self.out = out if out is not None else plan.new(u.cols)
        self.ws = torch.empty(L.ROWDOT_PARTS * 2 * self.P, dtype=torch.float32, device=plan.device)
        plan.add(self)

    def build_fwd(self, plan):
        r = self.r
        plan.fwd_steps.append(plan.call("cdc_cross_combine_fwd", self.x0.cptr(), C.c_int64(self.x0.ld), self.u.cptr(), C.c_int64(self.u.ld),
                                        _p(None if self.b1 is None else self.b1.data), _p(None if self.b2 is None else self.b2.data),
                                        None if r is None else r.cptr(), C.c_int64(0 if r is None else r.ld), self.out.cptr(),
                                        C.c_int64(self.out.ld), C.c_int64(plan.B), self.P, self.n_rep))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        for b in (self.x0, self.u, self.r):
            if b is not None and b.mask is not None:
                raise RuntimeError("cross_combine cannot consume an activation-fused linear output")
        og = self.out.grad
        plan.ensure_grad(self.x0, gs)                      # d_x0 accumulates inside the kernel
        x0g = self.x0.grad
        if gs.claim(self.u):
            raise RuntimeError("cross_combine: u feeds more than one consumer")
        ug = self.u.grad
        rg, acc_r = None, 0
        if self.r is not None:
            acc_r = 1 if gs.claim(self.r) else 0
            rg = self.r.grad
        db1 = db2 = None
        a1 = a2 = 0
        if self.b1 is not None:
            db1 = plan.param_grad(self.b1)
            a1 = 1 if plan._claim_param(self.b1) else 0
        if self.b2 is not None:
            db2 = plan.param_grad(self.b2)
            a2 = 1 if plan._claim_param(self.b2) else 0
        plan.bwd_steps.append(plan.call(
            "cdc_cross_combine_bwd", og.cptr(), C.c_int64(og.ld), self.x0.cptr(), C.c_int64(self.x0.ld), self.u.cptr(), C.c_int64(self.u.ld),
            _p(None if self.b1 is None else self.b1.data), ug.cptr(), C.c_int64(ug.ld), x0g.cptr(), C.c_int64(x0g.ld),
            None if rg is None else rg.cptr(), C.c_int64(0 if rg is None else rg.ld), acc_r, _p(db1), a1, _p(db2), a2, _p(self.ws),
            C.c_int64(plan.B), self.P, self.n_rep))


class AddOut:
    """out = a + b (CrossNetMix residual `moe_out + x_l`, model/layer.py:403)."""

    def __init__(self, plan, a, b, out=None):
        self.a, self.b = a, b
        self.out = out if out is not None else plan.new(a.cols)
        plan.add(self)

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_add_out", self.a.cptr(), C.c_int64(self.a.ld), self.b.cptr(), C.c_int64(self.b.ld),
                                        self.out.cptr(), C.c_int64(self.out.ld), C.c_int64(plan.B), self.a.cols))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        og = self.out.grad
        for t in (self.a, self.b):
            if t.mask is not None:
                raise RuntimeError("add cannot consume an activation-fused linear output")
            acc = 1 if gs.claim(t) else 0
            tg = t.grad
            plan.bwd_steps.append(plan.call("cdc_copy_or_add", tg.cptr(), C.c_int64(tg.ld), og.cptr(), C.c_int64(og.ld),
                                            C.c_int64(plan.B), t.cols, acc))


class StarFuse:
    """out_g = a_g (*|+) s for every domain g in one launch (model/star.py:90-93,100-102,169-176).
    a: list of per-domain Parameters, s: the shared Parameter; outputs live in one [n, ...] scratch tensor whose
    per-domain slices are handed to the consumers as TViews (their gradients come back through the same scratch)."""

    def __init__(self, plan, a_list, s, op):
        assert op in ("mul", "add")
        self.a_list, self.s, self.op = a_list, s, 0 if op == "mul" else 1
        n = len(a_list)
        shape = tuple(s.shape)
        self.out = torch.empty((n,) + shape, dtype=torch.float32, device=plan.device)
        self.d_out = torch.zeros((n,) + shape, dtype=torch.float32, device=plan.device)
        self.views = [TView(self.out[g], self.d_out[g]) for g in range(n)]
        self._keep = []
        plan.add(self)

    def _args(self, c0, chunk, backward, plan):
        a = L.StarFuseArgs()
        a.n, a.op, a.size = len(chunk), self.op, self.s.numel()
        a.s = self.s.data_ptr()
        if backward:
            a.ds = plan.param_grad(self.s).data_ptr()
            a.accumulate_ds = 1 if plan._claim_param(self.s) else 0
        for i, p in enumerate(chunk):
            a.a[i] = p.data_ptr()
            if backward:
                a.out[i] = self.d_out[c0 + i].data_ptr()
                a.da[i] = plan.param_grad(p).data_ptr()
                if plan._claim_param(p):
                    raise RuntimeError("a STAR domain parameter is used twice in one plan")
            else:
                a.out[i] = self.out[c0 + i].data_ptr()
        self._keep.append(a)
        return a

    def build_fwd(self, plan):
        for c0 in range(0, len(self.a_list), L.MAX_GROUPS):
            a = self._args(c0, self.a_list[c0:c0 + L.MAX_GROUPS], False, plan)
            plan.fwd_steps.append(plan.call("cdc_star_fuse_fwd", C.byref(a)))

    def build_bwd(self, plan, gs):
        for c0 in range(0, len(self.a_list), L.MAX_GROUPS):
            a = self._args(c0, self.a_list[c0:c0 + L.MAX_GROUPS], True, plan)
            plan.bwd_steps.append(plan.call("cdc_star_fuse_bwd", C.byref(a)))


class GroupPartition:
    """Stable partition of the batch rows by group id + permuted copy of a [B, C] buffer (model/star.py:84-86):
    rows come out in ascending group order, original order inside a group."""

    def __init__(self, plan, x, n_group):
        self.x, self.n_group = x, n_group
        self.group = torch.zeros(plan.B, dtype=torch.int64, device=plan.device)
        self.row_offsets = torch.zeros(n_group + 1, dtype=torch.int32, device=plan.device)
        self.order = torch.zeros(plan.B, dtype=torch.int32, device=plan.device)
        self.out = plan.new(x.cols)
        plan.add(self)

    def build_fwd(self, plan):
        plan.fwd_steps.append(plan.call("cdc_group_partition", _p(self.group), _p(self.row_offsets), _p(self.order), C.c_int64(plan.B),
                                        self.n_group))
        plan.fwd_steps.append(plan.call("cdc_rows_permute", self.x.cptr(), C.c_int64(self.x.ld), _p(self.order), self.out.cptr(),
                                        C.c_int64(self.out.ld), C.c_int64(plan.B), self.x.cols, 0))

    def build_bwd(self, plan, gs):
        plan.ensure_grad(self.out, gs)
        og = self.out.grad
        if gs.claim(self.x):
            tmp = plan.new(self.x.cols)
            xg = self.x.grad
            plan.bwd_steps.append(plan.call("cdc_rows_permute", og.cptr(), C.c_int64(og.ld), _p(self.order), tmp.cptr(), C.c_int64(tmp.ld),
                                            C.c_int64(plan.B), self.x.cols, 1))
            plan.bwd_steps.append(plan.call("cdc_copy_or_add", xg.cptr(), C.c_int64(xg.ld), tmp.cptr(), C.c_int64(tmp.ld),
                                            C.c_int64(plan.B), self.x.cols, 1))
        else:
            xg = self.x.grad
            plan.bwd_steps.append(plan.call("cdc_rows_permute", og.cptr(), C.c_int64(og.ld), _p(self.order), xg.cptr(), C.c_int64(xg.ld),
                                            C.c_int64(plan.B), self.x.cols, 1))
