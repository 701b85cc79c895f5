"""Dataset ingestion on device: the reference's pre-training host passes over a DSRL dataset --
``process_sequence_dataset`` (osrl/common/dataset.py:137-183), ``compute_cost_sample_prob`` (:439-459),
``process_bc_dataset`` (:30-134) -- run as HIP kernels (csrc/ingest.hip) on arrays uploaded once, and hand their
results straight to the on-device samplers (``SequenceStore`` / ``ReplayStore``) without a trip back to the host.

Same function names and argument meaning as the reference; inputs are the DSRL dict of numpy arrays (or device
tensors), outputs are device tensors.  There is no CPU path: a missing HIP library raises.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib as L
from ..engine.core import cur_stream, require_cuda

BC_MODES = {"all": 0, "multi-task": 0, "safe": 1, "risky": 2, "boundary": 3}  # include/osrl_amd.h OSRL_BC_*
COST_AFFINE, COST_RECIPROCAL = 0, 1


def _dev(x, device, dtype=torch.float32) -> torch.Tensor:
    t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=device, dtype=dtype).contiguous()


class Episodes:
    """Episode boundaries of a flat dataset, in HBM: ``start`` int64[n_ep], ``length`` int32[n_ep]."""

    def __init__(self, dataset: Dict[str, "np.ndarray | torch.Tensor"], device):
        self.device = require_cuda(device)
        lib = L.load()
        self.terminals, self.timeouts = _dev(dataset["terminals"], self.device), _dev(dataset["timeouts"], self.device)
        n = self.n = int(self.terminals.shape[0])
        self.ws = torch.zeros(int(lib.osrl_ingest_ws_elems(n)), dtype=torch.int32, device=self.device)
        end = torch.zeros(n, dtype=torch.int64, device=self.device)
        start = torch.zeros(n, dtype=torch.int64, device=self.device)
        length = torch.zeros(n, dtype=torch.int32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        L.check(lib.osrl_episode_segments(self.terminals.data_ptr(), self.timeouts.data_ptr(), n, end.data_ptr(),
                                          start.data_ptr(), length.data_ptr(), cnt.data_ptr(), self.ws.data_ptr(),
                                          cur_stream()), "osrl_episode_segments")
        self.n_episodes = int(cnt.item())  # the one host sync of ingestion: sizes the per-episode tables
        self.start, self.length = start[:self.n_episodes].clone(), length[:self.n_episodes].clone()
        # transitions covered by complete episodes (the tail after the last done flag is dropped / left at zero)
        self.n_covered = int((self.start[-1] + self.length[-1]).item()) if self.n_episodes else 0

    def returns(self, x: torch.Tensor, gamma: float, reverse: bool = False, broadcast_first: bool = False,
                x_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``discounted_cumsum`` (dataset.py:19-27) of ``x`` inside every episode (zeros outside)."""
        out = torch.zeros_like(x)
        L.check(L.load().osrl_episode_returns(x.data_ptr(), self.start.data_ptr(), self.length.data_ptr(),
                                              self.n_episodes, float(gamma), int(reverse), int(broadcast_first),
                                              out.data_ptr(), None if x_out is None else x_out.data_ptr(),
                                              cur_stream()), "osrl_episode_returns")
        return out


def process_sequence_dataset(dataset: Dict[str, "np.ndarray | torch.Tensor"], cost_reverse: bool = False,
                             device="cuda") -> Dict[str, torch.Tensor]:
    """dataset.py:137-183 on device.  Instead of a python list of per-episode dicts the result is the flat form the
    window sampler reads: ``observations [n,od]``, ``actions [n,ad]``, ``rewards``, ``costs`` (1 - c under
    ``cost_reverse``), ``returns`` / ``cost_returns`` (to-go sums, gamma = 1) over the n transitions covered by
    complete episodes, plus ``traj_start`` int64[n_traj] and ``traj_len`` int32[n_traj]."""
    ep = Episodes(dataset, device)
    n = ep.n_covered
    f = lambda k: _dev(dataset[k], ep.device)  # noqa: E731
    rew, cost_in = f("rewards"), f("costs")
    costs = torch.zeros_like(cost_in)
    cret = ep.returns(cost_in, 1.0, reverse=cost_reverse, x_out=costs)
    ret = ep.returns(rew, 1.0)
    obs, act = f("observations"), f("actions")
    return dict(observations=obs[:n], actions=act[:n], rewards=rew[:n], costs=costs[:n], returns=ret[:n],
                cost_returns=cret[:n], traj_start=ep.start, traj_len=ep.length)


CostTransform = Union[Tuple[str, float, float], Tuple[str, float]]


def compute_cost_sample_prob(tables: Dict[str, torch.Tensor], cost_transform: CostTransform = ("affine", -1.0, 50.0),
                             with_cdf: bool = False):
    """dataset.py:439-459 on device.  ``cost_transform`` names the two forms the reference's scripts use instead of a
    python callable: ``("affine", a, b)`` = ``a*x + b`` (the default ``50 - x``; train_cdt.py:139 ``70 - x``) or
    ``("reciprocal", b)`` = ``1 / (x + b)`` (train_cdt.py:139).  Returns prob (and the cdf the sampler reads)."""
    kind, a, b = (COST_AFFINE, float(cost_transform[1]), float(cost_transform[2])) if cost_transform[0] == "affine" \
        else (COST_RECIPROCAL, 0.0, float(cost_transform[1]))
    if cost_transform[0] not in ("affine", "reciprocal"):
        raise ValueError(cost_transform)
    n_traj = int(tables["traj_start"].shape[0])
    dev = tables["cost_returns"].device
    prob = torch.zeros(n_traj, dtype=torch.float32, device=dev)
    cdf = torch.zeros(n_traj, dtype=torch.float32, device=dev)
    L.check(L.load().osrl_cost_sample_prob(tables["cost_returns"].data_ptr(), tables["traj_start"].data_ptr(), n_traj,
                                           kind, a, b, prob.data_ptr(), cdf.data_ptr(), cur_stream()),
            "osrl_cost_sample_prob")
    return (prob, cdf) if with_cdf else prob


def compute_start_index_sample_prob(tables: Dict[str, torch.Tensor], prob: float = 0.4, with_cdf: bool = False):
    """dataset.py:472-494 on device: the per-trajectory start-index distribution of
    ``SequenceDataset(start_sampling=True)`` as ONE flat fp32 table [total rows] laid out like the trajectory tables
    (the reference returns a python list of per-trajectory fp64 arrays); ``with_cdf`` also returns the inclusive
    running sums inside each trajectory, which is what the window sampler reads."""
    costs = tables["costs"].contiguous()
    n_traj = int(tables["traj_start"].shape[0])
    p = torch.zeros_like(costs)
    cdf = torch.zeros_like(costs)
    L.check(L.load().osrl_start_index_prob(costs.data_ptr(), tables["traj_start"].data_ptr(),
                                           tables["traj_len"].data_ptr(), n_traj, float(prob), p.data_ptr(),
                                           cdf.data_ptr(), cur_stream()), "osrl_start_index_prob")
    return (p, cdf) if with_cdf else p


def process_bc_dataset(dataset: Dict[str, "np.ndarray | torch.Tensor"], cost_limit: float, gamma: float, bc_mode: str,
                       device="cuda") -> Dict[str, torch.Tensor]:
    """dataset.py:30-134 on device (all modes but "frontier", which needs oapackage's Pareto search): per-episode
    discounted returns broadcast to every transition, the mode's selection as a stable compaction, every array
    filtered, the cost return appended to the observation for "multi-task".  Returns NEW device tensors keyed like
    the input (plus ``cost_returns`` / ``rew_returns``); the reference edits its dict in place."""
    if bc_mode == "frontier":
        raise NotImplementedError('bc_mode="frontier" needs the oapackage Pareto search (not in this build)')
    if bc_mode not in BC_MODES:
        raise NotImplementedError(bc_mode)
    ep = Episodes(dataset, device)
    dev, n = ep.device, ep.n
    d = {k: _dev(v, dev) for k, v in dataset.items()}
    d["terminals"], d["timeouts"] = ep.terminals, ep.timeouts
    d["cost_returns"] = ep.returns(d["costs"], gamma, broadcast_first=True)
    d["rew_returns"] = ep.returns(d["rewards"], gamma, broadcast_first=True)
    f32 = lambda x: float(np.float32(x))  # noqa: E731  numpy compares the fp32 returns with fp32-rounded thresholds
    t0, t1 = {"safe": (f32(cost_limit), 0.0), "risky": (f32(2 * cost_limit), 0.0),
              "boundary": (f32(0.5 * cost_limit), f32(1.5 * cost_limit))}.get(bc_mode, (0.0, 0.0))
    idx = torch.zeros(n, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = L.load()
    L.check(lib.osrl_bc_select(d["cost_returns"].data_ptr(), n, BC_MODES[bc_mode], t0, t1, idx.data_ptr(),
                               cnt.data_ptr(), ep.ws.data_ptr(), cur_stream()), "osrl_bc_select")
    keep = int(cnt.item())
    out: Dict[str, torch.Tensor] = {}
    for k, v in d.items():
        w = int(v[0].numel()) if v.dim() > 1 else 1
        extra = d["cost_returns"] if (bc_mode == "multi-task" and k == "observations") else None
        cols = w + (1 if extra is not None else 0)
        dst = torch.zeros((keep, cols) if (v.dim() > 1 or extra is not None) else (keep,), dtype=torch.float32, device=dev)
        L.check(lib.osrl_gather_rows(v.data_ptr(), w, idx.data_ptr(), keep, dst.data_ptr(), cols,
                                     None if extra is None else extra.data_ptr(), cur_stream()), "osrl_gather_rows")
        out[k] = dst
    out["index"] = idx[:keep].clone()
    return out
