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
        # broadcasts name their source by GLOBAL rank: rank 0 of a sub-group is not global rank 0
        self.src0 = dist.get_global_rank(group, 0) if group is not None else 0
        self._gather_buf = None
        # bench.py's collective probe: a list makes every collective of the step bodies issued meanwhile record
        # (label, bytes, start event, end event) on its stream -- how long each one takes INSIDE the step
        self._probe: Optional[list] = None

    def _timed(self, label: str, nbytes: int, fn):
        if self._probe is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self._probe.append((label, int(nbytes), e0, e1))
        return r

    def rank_seed(self, seed: int) -> int:
        """Philox key of this rank's noise / dropout / sampling streams: the ranks' rows are different samples of one
        global batch, so their draws must be independent (same mixing as ReplayStore's per-rank seed)."""
        return int(seed) * 1000003 + self.rank

    # ---- collectives (plain torch.distributed; work on CPU tensors with gloo as well) ----
    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        self._timed("all_reduce", t.numel() * t.element_size(),
                    lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
        return t

    def all_reduce_many_(self, ts) -> None:
        """all-reduce(SUM) of several tensors as ONE collective launch (ncclGroupStart/End coalescing): the step's
        messages are 4 B - 1.6 MB, i.e. latency-bound on xGMI, so the count of collectives is what costs."""
        ts = [t for t in ts if t is not None]
        # (a subclass that supplies its own all_reduce_ -- the in-process stand-ins of tests/test_gpu_dp_sim.py -- must
        # not be bypassed by the coalesced RCCL path just because some process group happens to be initialised)
        own = type(self).all_reduce_ is DataParallel.all_reduce_
        if own and len(ts) > 1 and ts[0].is_cuda and dist.is_initialized() and dist.get_backend(self.group) == "nccl":
            def coalesced():
                with dist._coalescing_manager(group=self.group, device=ts[0].device, async_ops=False):
                    for t in ts:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self._timed(f"all_reduce x{len(ts)} coalesced", sum(t.numel() * t.element_size() for t in ts), coalesced)
        else:
            for t in ts:
                self.all_reduce_(t)

    def all_gather_concat(self, t: torch.Tensor) -> torch.Tensor:
        n = t.numel()
        if self._gather_buf is None or self._gather_buf.numel() != n * self.world or \
                self._gather_buf.device != t.device:
            self._gather_buf = torch.empty(n * self.world, dtype=t.dtype, device=t.device)
        self._timed("all_gather", n * t.element_size() * self.world,
                    lambda: dist.all_gather_into_tensor(self._gather_buf, t.reshape(-1), group=self.group))
        return self._gather_buf

    def all_agree(self, ok: bool, device) -> bool:
        """True iff EVERY rank passes ``ok`` (eager collective, outside any capture): used to settle graph replay vs
        eager launches for the whole group -- one rank replaying a graph with captured collectives while a peer
        issues them eagerly would still match up on the wire, but a rank that fell back must not be the only one.
        Scope: this settles capture-time REFUSALS (the runtime raising while the step is being captured, after which
        every rank arrives here).  A rank that raises from inside the warm-up pass BEFORE one of its collectives leaves
        its peers blocked in that collective -- like any mid-step failure of a data-parallel job; the launcher's
        timeout (torchrun / NCCL_TIMEOUT) is what ends it, not this vote."""
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=device)
        self.all_reduce_(t)
        return bool(float(t.item()) >= self.world - 0.5)

    def broadcast_(self, t: torch.Tensor) -> torch.Tensor:
        dist.broadcast(t, src=self.src0, group=self.group)
        return t

    def broadcast_model(self, model, engine=None) -> None:
        """Make every replica bit-identical to the group's rank 0: parameters, targets, optimizer moments, scalar
        state and -- with ``engine`` -- the train-step counter (Adam bias correction, warm-up LR, Philox offsets) and
        CDT's temperature moments.  The step engines call this from their constructor hand-off
        (common/checkpoint.py ``engine_handoff``), so a data-parallel engine never starts from diverged replicas."""
        for g in model.groups.values():
            for buf in (g.p, g.m, g.v, g.tgt):
                if buf is not None:
                    self.broadcast_(buf)
        for name in ("log_alpha", "pid_state", "log_temperature", "scalar_leaves"):
            if isinstance(getattr(model, name, None), torch.Tensor):
                self.broadcast_(getattr(model, name))
        if engine is not None:
            step = torch.tensor([engine.st.device_step()], dtype=torch.int64, device=engine.st.state.device)
            self.broadcast_(step)
            engine.st.set_step(int(step.item()))
            if getattr(engine, "temp_mv", None) is not None:
                self.broadcast_(engine.temp_mv)
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
        self.quantile_select(self.all_gather_concat(local_vals), q, out)

    def quantile_select(self, allv: torch.Tensor, q: float, out: torch.Tensor) -> None:
        """The q-quantile of the gathered values (no collective inside: may run on a side stream)."""
        from . import glue as G
        n = allv.numel()
        if n > 32768:  # past the register-resident single-workgroup select: the grid version (200 -> ~30 us at 8 ranks)
            ws = getattr(self, "_qws", None)
            if ws is None or ws.device != allv.device:
                ws = self._qws = torch.zeros(L.QUANTILE_WS, dtype=torch.int32, device=allv.device)
            G.quantile_ws(allv, n, q, ws, out)
        else:
            G.quantile(allv, n, q, out)
