"""Device-resident transition store + on-device uniform sampler.

Replaces the reference's minibatch source -- ``TransitionDataset`` (osrl/common/dataset.py:790-847:
``done = terminals | timeouts`` as fp32 :815-816, ``rewards*reward_scale``, ``costs*cost_scale``,
one uniform-with-replacement index per sample :846) + torch ``DataLoader`` + six ``.to(device)``
copies per step (examples/train/train_cpq.py:122-142) -- with tables that stay in HBM and a gather
kernel (csrc/rng.hip ``osrl_replay_gather``) that draws the indices on device inside the captured
train step.  In the data-parallel setting every rank holds its own shard of the transitions
(``shard(rank, world)``) and samples locally: with random partitions this is distributionally the
same as global uniform sampling.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .. import _lib as L
from ..engine.core import cur_stream

FIELDS = ("observations", "next_observations", "actions", "rewards", "costs", "done")


class ReplayStore:
    def __init__(self, data: Dict[str, "np.ndarray | torch.Tensor"], device, reward_scale: float = 1.0,
                 cost_scale: float = 1.0, seed: int = 0, rank: int = 0, world: int = 1, state_init: bool = False):
        """``data`` uses the DSRL dataset keys (observations, next_observations, actions, rewards, costs,
        and either ``done`` or ``terminals``+``timeouts``).  ``state_init`` (TransitionDataset(state_init=True),
        dataset.py:817-820, used by COptiDICE): a 7th table ``is_init`` = ``done`` shifted by one transition with
        ``is_init[0] = 1``, computed on the FULL dataset before any sharding."""
        d = dict(data)
        self.state_init = bool(state_init)
        if "done" not in d:
            if torch.is_tensor(d["terminals"]):  # device tables of common.ingest.process_bc_dataset
                d["done"] = torch.logical_or(d["terminals"] == 1, d["timeouts"] == 1)
            else:
                d["done"] = np.logical_or(np.asarray(d["terminals"]) == 1, np.asarray(d["timeouts"]) == 1)
        n = len(d["observations"])
        fields = FIELDS
        if self.state_init:
            dn = d["done"]
            dn = dn.detach().cpu().numpy() if torch.is_tensor(dn) else np.asarray(dn)
            init = np.asarray(dn, np.float32).reshape(-1).copy()
            init[1:] = init[:-1]
            init[0] = 1.0
            d["is_init"] = init
            fields = FIELDS + ("is_init",)
            # TransitionDataset.get_dataset_states (dataset.py:822-830): what COptiDICE's constructor is fed
            as_np = lambda x: x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)  # noqa: E731
            self.init_state_propotion = float(init.mean())
            self.observations_std = as_np(d["observations"]).std(0, keepdims=True)
            self.actions_std = as_np(d["actions"]).std(0, keepdims=True)
        sl = slice(rank, n, world) if world > 1 else slice(None)
        self.tables = []
        for k in fields:
            t = torch.as_tensor(np.asarray(d[k])[sl] if not torch.is_tensor(d[k]) else d[k][sl])
            t = t.to(device=device, dtype=torch.float32).reshape(t.shape[0], -1).contiguous()
            self.tables.append(t)
        self.n_rows = self.tables[0].shape[0]
        self.widths = [t.shape[1] for t in self.tables]
        self.scales = [1.0, 1.0, 1.0, float(reward_scale), float(cost_scale), 1.0] + ([1.0] if self.state_init else [])
        self.seed = int(seed) * 1000003 + rank
        self.device = torch.device(device)
        nf = self.n_fields = len(self.tables)
        self._src = (C.c_void_p * nf)(*[t.data_ptr() for t in self.tables])
        self._w = (C.c_int32 * nf)(*self.widths)
        self._s = (C.c_float * nf)(*self.scales)
        self.bytes_per_row = 4 * sum(self.widths)

    def get_dataset_states(self):
        """(init_state_propotion, observations_std, actions_std) -- dataset.py:822-830; needs ``state_init``."""
        if not self.state_init:
            raise RuntimeError("build the store with state_init=True")
        return self.init_state_propotion, self.observations_std, self.actions_std

    def gather(self, dst: Sequence[torch.Tensor], st_ptr: Optional[int], idx_out: Optional[torch.Tensor] = None,
               stream_id: int = 1) -> None:
        """dst = (obs, next_obs, act, rew, cost, done[, is_init]) batch buffers; asynchronous on the current stream."""
        B = dst[0].shape[0]
        if len(dst) != self.n_fields:
            raise ValueError(f"the store holds {self.n_fields} tables, {len(dst)} destination buffers were given")
        d = (C.c_void_p * self.n_fields)(*[t.data_ptr() for t in dst])
        L.check(L.load().osrl_replay_gather(self.n_fields, self._src, d, self._w, self._s, self.n_rows, B,
                                            None if idx_out is None else idx_out.data_ptr(), self.seed, stream_id,
                                            st_ptr, cur_stream()), "osrl_replay_gather")


    def gather_args(self, dst: Sequence[torch.Tensor], fields: Optional[Sequence[int]] = None, stream_id: int = 1):
        """The arguments of ``gather`` / ``gather_fields`` as the tuple ``StepState.begin(gather=...)`` takes (the
        fused step prologue draws the same rows: the indices are a function of (seed, step, row) only)."""
        if fields is None:
            fields = range(self.n_fields)
        fields = list(fields)
        if len(dst) != len(fields):
            raise ValueError(f"{len(fields)} tables selected, {len(dst)} destination buffers were given")
        n = len(fields)
        src = (C.c_void_p * n)(*[self.tables[i].data_ptr() for i in fields])
        d = (C.c_void_p * n)(*[t.data_ptr() for t in dst])
        w = (C.c_int32 * n)(*[self.widths[i] for i in fields])
        sc = (C.c_float * n)(*[self.scales[i] for i in fields])
        return (n, src, d, w, sc, self.n_rows, dst[0].shape[0], self.seed, stream_id, list(dst))

    def gather_fields(self, fields: Sequence[int], dst: Sequence[torch.Tensor], st_ptr: Optional[int],
                      stream_id: int = 1) -> None:
        """The same draw as ``gather`` (the row indices are a function of (seed, step, row) only) restricted to some
        of the tables -- (0, 2) = observations, actions is all BC reads (train_bc.py:121)."""
        n, B = len(fields), dst[0].shape[0]
        src = (C.c_void_p * n)(*[self.tables[i].data_ptr() for i in fields])
        d = (C.c_void_p * n)(*[t.data_ptr() for t in dst])
        w = (C.c_int32 * n)(*[self.widths[i] for i in fields])
        sc = (C.c_float * n)(*[self.scales[i] for i in fields])
        L.check(L.load().osrl_replay_gather(n, src, d, w, sc, self.n_rows, B, None, self.seed, stream_id, st_ptr,
                                            cur_stream()), "osrl_replay_gather")


def synthetic_transitions(n: int, od: int, ad: int, seed: int = 1, max_action: float = 1.0) -> Dict[str, np.ndarray]:
    """Synthetic DSRL-shaped data: obs~N(0,1), act~U(-1,1), rew~N(0,1), cost~Bern(.1),
    terminals~Bern(.01)."""
    rs = np.random.RandomState(seed)
    f = np.float32
    return dict(observations=rs.randn(n, od).astype(f), next_observations=rs.randn(n, od).astype(f),
                actions=(rs.uniform(-1, 1, (n, ad)) * max_action).astype(f), rewards=rs.randn(n).astype(f),
                costs=(rs.uniform(size=n) < 0.1).astype(f), terminals=(rs.uniform(size=n) < 0.01).astype(f),
                timeouts=np.zeros(n, f))


class SequenceStore:
    """Device-resident trajectory store + on-device window sampler for CDT: replaces ``SequenceDataset``
    (osrl/common/dataset.py:633-787; augmentation / Pareto constructor paths are out of scope) + DataLoader.
    ``trajectories``: list of dicts with observations [L,od], actions [L,ad], returns [L] (return-to-go),
    cost_returns [L] (cost-to-go), costs [L].  ``sample_prob``: optional per-trajectory probabilities
    (dataset.py:439-459 ``compute_cost_sample_prob`` output); None = uniform."""

    def __init__(self, trajectories, seq_len: int, device, reward_scale: float = 1.0, cost_scale: float = 1.0,
                 sample_prob=None, seed: int = 0, rank: int = 0, start_sampling: bool = False, prob: float = 0.4):
        self.T, self.device = int(seq_len), torch.device(device)
        cat = lambda k: torch.as_tensor(np.concatenate([np.asarray(t[k], np.float32).reshape(len(t["costs"]), -1)  # noqa: E731
                                                        for t in trajectories]), device=self.device).contiguous()
        self.obs, self.act = cat("observations"), cat("actions")
        self.ret, self.cret, self.cost = cat("returns").view(-1), cat("cost_returns").view(-1), cat("costs").view(-1)
        lens = np.array([len(t["costs"]) for t in trajectories], np.int64)
        self.traj_len = torch.as_tensor(lens.astype(np.int32), device=self.device)
        self.traj_start = torch.as_tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64), device=self.device)
        self.n_traj = len(trajectories)
        self.cdf = None
        if sample_prob is not None:
            c = np.cumsum(np.asarray(sample_prob, np.float64))
            c /= c[-1]
            self.cdf = torch.as_tensor(c.astype(np.float32), device=self.device)
        self.reward_scale, self.cost_scale = float(reward_scale), float(cost_scale)
        self.base_seed = int(seed)
        self.set_rank(rank)
        self.od, self.ad = self.obs.shape[1], self.act.shape[1]
        self.start_cdf = None
        if start_sampling:
            self.enable_start_sampling(prob)

    def enable_start_sampling(self, prob: float = 0.4) -> None:
        """``SequenceDataset(start_sampling=True, prob=...)`` (dataset.py:742-744,781-783): window starts are drawn
        from ``compute_start_index_sample_prob`` instead of uniformly (computed on device, csrc/ingest.hip)."""
        from .ingest import compute_start_index_sample_prob
        tables = dict(costs=self.cost, traj_start=self.traj_start, traj_len=self.traj_len)
        self.start_cdf = compute_start_index_sample_prob(tables, prob, with_cdf=True)[1]

    def set_rank(self, rank: int) -> None:
        """Data parallel: every rank draws its own windows (same mixing as ReplayStore); the CDT engine calls this
        from ``attach_store`` when it runs under a DataParallel hook."""
        self.rank = int(rank)
        self.seed = self.base_seed * 1000003 + self.rank if self.rank else self.base_seed

    @classmethod
    def from_tables(cls, tables: Dict[str, torch.Tensor], seq_len: int, reward_scale: float = 1.0,
                    cost_scale: float = 1.0, cdf: Optional[torch.Tensor] = None, seed: int = 0,
                    rank: int = 0, start_sampling: bool = False, prob: float = 0.4) -> "SequenceStore":
        """Wrap the flat device tables of ``common.ingest.process_sequence_dataset`` (nothing is copied)."""
        self = cls.__new__(cls)
        self.T, self.device = int(seq_len), tables["observations"].device
        self.obs, self.act = tables["observations"].contiguous(), tables["actions"].contiguous()
        self.ret, self.cret, self.cost = tables["returns"], tables["cost_returns"], tables["costs"]
        self.traj_start, self.traj_len = tables["traj_start"], tables["traj_len"]
        self.n_traj = int(self.traj_start.shape[0])
        self.cdf = cdf
        self.reward_scale, self.cost_scale = float(reward_scale), float(cost_scale)
        self.base_seed = int(seed)
        self.set_rank(rank)
        self.od, self.ad = self.obs.shape[1], self.act.shape[1]
        self.start_cdf = None
        if start_sampling:
            self.enable_start_sampling(prob)
        return self

    @classmethod
    def from_dataset(cls, dataset, seq_len: int, device, reward_scale: float = 1.0, cost_scale: float = 1.0,
                     cost_reverse: bool = False, cost_sample: bool = False,
                     cost_transform=("affine", -1.0, 50.0), seed: int = 0, rank: int = 0,
                     start_sampling: bool = False, prob: float = 0.4) -> "SequenceStore":
        """``SequenceDataset(dataset, seq_len, reward_scale, cost_scale, cost_reverse=, cost_sample=,
        cost_transform=)`` (dataset.py:633-741, no augmentation) with the whole preprocessing on device."""
        from .ingest import compute_cost_sample_prob, process_sequence_dataset
        tables = process_sequence_dataset(dataset, cost_reverse, device)
        cdf = compute_cost_sample_prob(tables, cost_transform, with_cdf=True)[1] if cost_sample else None
        return cls.from_tables(tables, seq_len, reward_scale, cost_scale, cdf, seed, rank, start_sampling, prob)

    def gather(self, states, actions, returns, cost_returns, time_steps, mask, episode_cost, costs, st_ptr,
               idx_out=None, stream_id: int = 2, idx_in=None) -> None:
        """``idx_in``: optional int32 [B,2] device tensor of (trajectory, start) pairs to use instead of drawing."""
        B = states.shape[0]
        L.check(L.load().osrl_seq_window_gather(
            self.obs.data_ptr(), self.act.data_ptr(), self.ret.data_ptr(), self.cret.data_ptr(), self.cost.data_ptr(),
            self.traj_start.data_ptr(), self.traj_len.data_ptr(), None if self.cdf is None else self.cdf.data_ptr(),
            None if self.start_cdf is None else self.start_cdf.data_ptr(),
            None if idx_in is None else idx_in.data_ptr(), self.n_traj, B, self.T, self.od, self.ad, self.reward_scale, self.cost_scale, states.data_ptr(),
            actions.data_ptr(), returns.data_ptr(), cost_returns.data_ptr(), time_steps.data_ptr(), mask.data_ptr(),
            episode_cost.data_ptr(), costs.data_ptr(), None if idx_out is None else idx_out.data_ptr(), self.seed,
            stream_id, st_ptr, cur_stream()), "osrl_seq_window_gather")
