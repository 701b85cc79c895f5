"""Data-parallel wiring: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference has no distributed code at all; this is new design.  Rows of a
minibatch are independent in every forward/backward, so each rank runs the same step plan on its own
B-row shard (drawn from its own shard of the replay store) with every 1/B normalisation using the
GLOBAL batch, and exchanges exactly what a single device would have reduced over the batch:

  * per optimizer phase: the flat gradient of that group -- ONE all-reduce(SUM) of 0.3-1.6 MB right
    before the group's fused Adam kernel (message-latency bound on xGMI, so one bucket per phase);
  * CPQ: the N*B KL values for the batch-global 0.75-quantile (all-gather, 80 KB/rank) and the scalar
    mean of qc_ood that drives the log_alpha ascent;
  * logged statistics: the <=8-float stats vector at the end of the step.
  Everything is latency-bound, so CPQ coalesces what has no dependency in between into one launch
  (``all_reduce_many_``): [critic grads | cost-critic grads | qc_ood mean] and [actor grads | statistics] --
  4 collectives per step (VAE grads, KL gather, those two) instead of 7.

Oracle for correctness: a sharded step on W x B rows == the single-device step on the concatenated
batch (tests/test_dist_cpu.py checks the reduction algebra with gloo, world_size 2, on CPU).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .. import _lib as L
from .core import FlatGroup, cur_stream


class DataParallel:
    def __init__(self, group: Optional["dist.ProcessGroup"] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._gather_buf = None

    # ---- collectives (plain torch.distributed; work on CPU tensors with gloo as well) ----
    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_many_(self, ts) -> None:
        """all-reduce(SUM) of several tensors as ONE collective launch (ncclGroupStart/End coalescing): the step's
        messages are 4 B - 1.6 MB, i.e. latency-bound on xGMI, so the count of collectives is what costs."""
        ts = [t for t in ts if t is not None]
        if len(ts) > 1 and ts[0].is_cuda and dist.is_initialized() and dist.get_backend(self.group) == "nccl":
            with dist._coalescing_manager(group=self.group, device=ts[0].device, async_ops=False):
                for t in ts:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            for t in ts:
                self.all_reduce_(t)

    def all_gather_concat(self, t: torch.Tensor) -> torch.Tensor:
        n = t.numel()
        if self._gather_buf is None or self._gather_buf.numel() != n * self.world or \
                self._gather_buf.device != t.device:
            self._gather_buf = torch.empty(n * self.world, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(self._gather_buf, t.reshape(-1), group=self.group)
        return self._gather_buf

    def broadcast_model(self, model) -> None:
        """Make every replica bit-identical to rank 0 (parameters, targets, optimizer moments, scalars)."""
        for g in model.groups.values():
            for buf in (g.p, g.m, g.v, g.tgt):
                if buf is not None:
                    dist.broadcast(buf, src=0, group=self.group)
        for name in ("log_alpha", "pid_state", "log_temperature", "scalar_leaves"):
            if isinstance(getattr(model, name, None), torch.Tensor):
                dist.broadcast(getattr(model, name), src=0, group=self.group)
        model.repack()  # the kernels read fragment-ordered copies of the weights: refresh them from the new values

    # ---- hooks used by the step engines ----
    def reduce_local(self, grp: FlatGroup) -> torch.Tensor:
        """Sum the split-K slabs of this rank into slab 0 (HIP kernel) and return it."""
        if grp.cur_splits > 1:
            L.check(L.load().osrl_reduce_slabs(grp.slabs.data_ptr(), grp.slabs.data_ptr(), grp.cur_splits, grp.n,
                                               grp.n, cur_stream()), "osrl_reduce_slabs")
            grp.cur_splits = 1
        return grp.slabs[0]

    def allreduce_group(self, grp: FlatGroup) -> None:
        self.all_reduce_(self.reduce_local(grp))

    def quantile(self, local_vals: torch.Tensor, q: float, out: torch.Tensor) -> None:
        from . import glue as G
        allv = self.all_gather_concat(local_vals)
        G.quantile(allv, allv.numel(), q, out)
