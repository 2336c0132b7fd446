"""Data parallelism for the training step: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm;
"gloo" for the CPU rehearsal of the same call pattern in tests).

The reference has no multi-device code at all; the exchange pattern is this build's:
  * samples shard across ranks (global batch = world_size x local batch, BCE mean over the GLOBAL batch);
  * dense gradients: ONE sum all-reduce of the flat gradient arena per step;
  * embedding table, "sharded" (default with the lazy table optimiser): row r is owned by rank r % world_size; per step
    three equal-split all-to-alls move (row ids -> owners), (rows -> requesters), (row gradients -> owners); the owner
    alone replays / updates its rows.  Per-rank table work and traffic are those of the LOCAL batch, whatever the world size;
  * embedding table, "replicated": all-gather of the (row index, row gradient) pairs, then every rank applies the
    identical table update — no divergence, no table broadcast, but table work grows with the global batch.
"""
import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, backend=None, device=None, force=None):
        """force (default: env CDC_FORCE_COLLECTIVES=1): create the process group and issue every collective even when the
        world has ONE rank — the way to run the real RCCL call path (communicator streams, async handles, graph segments
        between collectives) on a single-GPU machine; results equal the single-process step."""
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.force = bool(int(os.environ.get("CDC_FORCE_COLLECTIVES", "0"))) if force is None else bool(force)
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        self.backend = backend
        self.active = self.world_size > 1 or self.force
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend, device_id=torch.device("cuda", self.local_rank), rank=self.rank,
                                        world_size=self.world_size)
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world_size)
        self.device = device
        self._on_close = []               # callbacks run before the process group goes (graphs that captured collectives must go first)

    def _staged(self, t):
        """gloo moves host memory: device tensors are staged through the CPU (rehearsal / tests only; RCCL is direct)."""
        return self.backend == "gloo" and t.is_cuda

    def all_reduce_sum(self, t):
        if self.active:
            if self._staged(t):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_reduce_max(self, t):
        if self.active:
            if self._staged(t):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    def all_gather_rows(self, out, local):
        """out [world*B, C] <- concat over ranks of local [B, C] (rank order)."""
        if self.active:
            if self._staged(out):
                parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(self.world_size)]
                dist.all_gather(parts, local.detach().cpu().contiguous())
                out.copy_(torch.cat(parts, dim=0))
            elif self.backend == "gloo":
                parts = list(out.chunk(self.world_size, dim=0))
                dist.all_gather(parts, local.contiguous())
            else:
                dist.all_gather_into_tensor(out, local.contiguous())
        else:
            out.copy_(local)
        return out

    def all_to_all(self, out, inp):
        """equal-split exchange along dim 0: chunk r of `inp` goes to rank r; chunk r of `out` came from rank r."""
        if self.active:
            if self._staged(out):
                h_in = inp.detach().cpu().contiguous()
                h_out = torch.empty_like(h_in)
                dist.all_to_all_single(h_out, h_in)
                out.copy_(h_out)
            else:
                dist.all_to_all_single(out, inp)
        else:
            out.copy_(inp)
        return out

    def all_to_all_start(self, out, inp):
        """all_to_all whose completion the caller awaits later with wait(): over RCCL the exchange runs on the
        communicator's stream while the launches issued in between run on the compute stream.  Host-staged backends (the
        gloo rehearsal) complete at once and return None."""
        if self.active and not self._staged(out) and self.backend != "gloo":
            return dist.all_to_all_single(out, inp, async_op=True)
        self.all_to_all(out, inp)
        return None

    @staticmethod
    def wait(handle):
        if handle is not None:
            handle.wait()                      # orders the current stream after the collective; does not block the host

    def barrier(self):
        if self.active:
            dist.barrier()

    def shard(self, n):
        """contiguous [begin, end) of n items owned by this rank."""
        per = n // self.world_size
        return self.rank * per, (self.rank + 1) * per

    @property
    def capturable(self):
        """collectives of this backend can be captured into a hipGraph together with the launches around them (RCCL: yes — round 4
        measured capture and replay of all-reduce / all-to-all on this stack; the host-staged gloo rehearsal: no)"""
        return self.active and self.backend == "nccl"

    def close(self):
        if self.active and dist.is_initialized():
            # a live graph that captured a collective makes destroy_process_group() hang (PyTorch 2.10 / RCCL of ROCm 7: the
            # "capture never returns" of rounds 2-3 was this): release them first
            for fn in self._on_close:
                fn()
            self._on_close.clear()
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            dist.destroy_process_group()
