"""Data parallelism WITHOUT collective-library launches: the step's exchanges through IPC-mapped device buffers
(csrc/ipc.hip).

``IpcDataParallel`` is a drop-in for ``DataParallel`` (engine/dist.py): the engines call the same three primitives --
``all_reduce_``, ``all_reduce_many_``, ``all_gather_concat`` -- and each becomes ONE kernel launch per rank that publishes
the local values in the rank's own buffer, raises a flag, waits for every peer's flag and sums (in rank order: replicas
stay bit-identical) or gathers what the peers published.  ``torch.distributed`` -- ANY backend -- is used once, at
construction, to hand the IPC handles around, and for the setup-time broadcast of rank 0's replica; nothing of it runs
inside a step.  Because the backend does not matter, two PROCESSES can share one GPU (gloo as the control plane): that is
how the path is exercised with a real peer on a one-GPU box (tests/test_gpu_ipc_dp.py), which RCCL refuses ("Duplicate
GPU detected").  On a multi-GPU node the same handles map peer memory over xGMI (hipIpcMemLazyEnablePeerAccess); that has
not been run -- select it with ``OSRL_DP_EXCHANGE=ipc`` (bench.py), the default stays RCCL.

Reference: none -- the reference has no distributed code; what is exchanged and why is SURVEY.md 8e.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _lib as L
from .core import cur_stream
from .dist import DataParallel


class IpcDataParallel(DataParallel):
    def __init__(self, group: Optional["dist.ProcessGroup"] = None, half_floats: int = 1 << 20, device=None):
        """``half_floats``: capacity of one published half (4 MB by default: the largest message of the step plans is the
        1.56 MB VAE gradient at C2, [critic | cost critic] 1.4 MB); a larger message raises."""
        super().__init__(group)
        if self.world > L.IPC_MAX_WORLD:
            raise RuntimeError(f"IpcDataParallel: world {self.world} > {L.IPC_MAX_WORLD}")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        lib = L.load()
        self.half = (int(half_floats) + 3) & ~3
        pub, ctl = C.c_void_p(), C.c_void_p()
        hp, hc = (C.c_char * L.IPC_HANDLE_BYTES)(), (C.c_char * L.IPC_HANDLE_BYTES)()
        with torch.cuda.device(self.device):
            L.check(lib.osrl_ipc_alloc(2 * self.half * 4, C.byref(pub), hp), "osrl_ipc_alloc(pub)")
            L.check(lib.osrl_ipc_alloc(64, C.byref(ctl), hc), "osrl_ipc_alloc(ctl)")
        self._own = (pub.value, ctl.value)
        mine = (bytes(hp), bytes(hc))
        everyone: List = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        x = L.IpcT()
        x.world, x.rank, x.half_floats = self.world, self.rank, self.half
        self._mapped = []
        with torch.cuda.device(self.device):
            for r, (bp, bc) in enumerate(everyone):
                if r == self.rank:
                    x.pub[r], x.ctl[r] = pub.value, ctl.value
                    continue
                pp, pc = C.c_void_p(), C.c_void_p()
                L.check(lib.osrl_ipc_open(C.create_string_buffer(bp, L.IPC_HANDLE_BYTES), C.byref(pp)), "osrl_ipc_open(pub)")
                L.check(lib.osrl_ipc_open(C.create_string_buffer(bc, L.IPC_HANDLE_BYTES), C.byref(pc)), "osrl_ipc_open(ctl)")
                x.pub[r], x.ctl[r] = pp.value, pc.value
                self._mapped += [pp.value, pc.value]
        self.x = x
        self._pending = {}
        dist.barrier(group=group)  # every rank has mapped every buffer before the first exchange can publish into one

    # ---- the rank's own split-K slab sum rides on the exchange ----
    FUSE_SLABS = os.environ.get("OSRL_IPC_FUSE_SLABS", "1") == "1"  # (0: reduce_local launches the slab sum; the test compares)

    def reduce_local(self, grp) -> torch.Tensor:
        """``DataParallel.reduce_local`` launches the rank's slab sum; here it is only NOTED: the exchange that follows
        forms it while publishing (csrc/ipc.hip publish_seg: same slab order, same bits) and leaves the sum over slabs and
        ranks in slab 0.  The engines hand the returned tensor straight to ``all_reduce_`` / ``all_reduce_many_``."""
        if not self.FUSE_SLABS or grp.cur_splits <= 1:
            return super().reduce_local(grp)
        self._pending[grp.slabs.data_ptr()] = (int(grp.cur_splits), int(grp.n))
        grp.cur_splits = 1
        return grp.slabs[0]

    # ---- the three primitives the engines use ----
    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("IpcDataParallel exchanges contiguous fp32 device tensors")
        return t

    def all_reduce_many_(self, ts) -> None:
        ts = [self._f32(t) for t in ts if t is not None]
        i = 0
        while i < len(ts):  # as many tensors per launch as fit a half (and the segment table)
            j, tot = i, 0
            while j < len(ts) and j - i < L.IPC_MAX_SEG and tot + ((ts[j].numel() + 3) & ~3) <= self.half:
                tot += (ts[j].numel() + 3) & ~3
                j += 1
            if j == i:
                raise RuntimeError(f"IpcDataParallel: a {ts[i].numel()}-float message does not fit a published half of "
                                   f"{self.half} floats (construct with a larger half_floats)")
            chunk = ts[i:j]
            bufs = (C.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
            lens = (C.c_int64 * len(chunk))(*[t.numel() for t in chunk])
            pend = [self._pending.pop(t.data_ptr(), (1, 0)) for t in chunk]  # (slab 0 of a group whose local sum is due)
            nsl = (C.c_int32 * len(chunk))(*[p[0] for p in pend])
            sst = (C.c_int64 * len(chunk))(*[p[1] for p in pend])
            self._timed(f"ipc all_reduce x{len(chunk)}", 4 * sum(t.numel() for t in chunk),
                        lambda: L.check(L.load().osrl_ipc_all_reduce_slabs(C.byref(self.x), bufs, lens, nsl, sst, len(chunk),
                                                                           cur_stream()), "osrl_ipc_all_reduce_slabs"))
            i = j

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        self.all_reduce_many_([t])
        return t

    def all_gather_concat(self, t: torch.Tensor) -> torch.Tensor:
        t = self._f32(t)
        n = t.numel()
        if n > self.half:
            raise RuntimeError(f"IpcDataParallel: a {n}-float gather does not fit a published half of {self.half} floats")
        if self._gather_buf is None or self._gather_buf.numel() != n * self.world or self._gather_buf.device != t.device:
            self._gather_buf = torch.empty(n * self.world, dtype=t.dtype, device=t.device)
        self._timed("ipc all_gather", 4 * n * self.world,
                    lambda: L.check(L.load().osrl_ipc_all_gather(C.byref(self.x), t.data_ptr(), n,
                                                                 self._gather_buf.data_ptr(), cur_stream()),
                                    "osrl_ipc_all_gather"))
        return self._gather_buf

    # ---- health ----
    def status(self) -> dict:
        """This rank's control words (synchronises): ``error`` 0 = fine, 1 + r = rank r never published an exchange this rank
        waited for (its launch gave up after ~2 s and left the destination unreduced)."""
        w = (C.c_uint32 * 4)()
        with torch.cuda.device(self.device):
            L.check(L.load().osrl_ipc_status(C.byref(self.x), w), "osrl_ipc_status")
        return {"flag": int(w[0]), "arrivals": int(w[1]), "error": int(w[2]), "done": int(w[3])}

    def check(self) -> None:
        s = self.status()
        if s["error"]:
            raise RuntimeError(f"osrl_amd: rank {self.rank} gave up waiting for rank {s['error'] - 1} in exchange "
                               f"{s['flag']}: the replicas have diverged -- restore a checkpoint")

    def close(self) -> None:
        """Unmap the peers' buffers and free this rank's (after a barrier: nobody may still read them)."""
        if getattr(self, "x", None) is None:
            return
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        lib = L.load()
        with torch.cuda.device(self.device):
            for p in self._mapped:
                lib.osrl_ipc_close(p)
            dist.barrier(group=self.group)
            for p in self._own:
                lib.osrl_ipc_free(p)
        self.x = None
