"""CPU oracle for dataset ingestion (osrl/common/dataset.py of the reference): episode segmentation,
return-to-go / cost-to-go, BC trajectory filters, CDT cost-weighted trajectory sampling probabilities.

TEST INFRASTRUCTURE ONLY (same rules as osrl_oracle.py): nothing under ``osrl_amd/`` may import it.
numpy restatement; every function cites the reference file:line it follows.  PINNED by
``tests/golden/ingest.npz`` / ``samples.npz`` -- outputs of the reference's own functions, captured by importing the
reference (``tests/golden/make_golden_ingest.py``) -- see ``tests/test_oracle_golden.py::test_ingest_oracle_matches_golden``
and ``::test_minibatch_builders_match_reference_samples``.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

Array = np.ndarray


def done_flags(dataset: Dict[str, Array]) -> Array:
    """An episode ends where ``terminals`` or ``timeouts`` is set (dataset.py:60, :165)."""
    return np.logical_or(np.asarray(dataset["terminals"]) == 1, np.asarray(dataset["timeouts"]) == 1)


def episode_segments(done: Array) -> Tuple[Array, Array]:
    """(start, length) of every COMPLETE episode.  Transitions after the last done flag belong to no episode:
    process_sequence_dataset never flushes its running buffer for them (dataset.py:156-175) and
    process_bc_dataset's loop runs over ``done_idx`` only (dataset.py:60-70)."""
    ends = np.flatnonzero(done)
    starts = np.concatenate([[0], ends[:-1] + 1]) if ends.size else np.zeros(0, np.int64)
    return starts.astype(np.int64), (ends - starts + 1).astype(np.int64)


def discounted_cumsum(x: Array, gamma: float) -> Array:
    """dataset.py:19-27: c[T-1] = x[T-1]; c[t] = x[t] + gamma * c[t+1], evaluated in x's dtype (fp32 data stay
    fp32: one rounded multiply and one rounded add per element, back to front)."""
    x = np.asarray(x)
    g = x.dtype.type(gamma)
    c = np.empty_like(x)
    acc = x.dtype.type(0)
    for t in range(x.shape[0] - 1, -1, -1):
        acc = x[t] + g * acc if t < x.shape[0] - 1 else x[t]
        c[t] = acc
    return c


def process_sequence_dataset(dataset: Dict[str, Array], cost_reverse: bool = False) -> List[Dict[str, Array]]:
    """dataset.py:137-183: split at done flags; per episode fp32 observations/actions/rewards/costs (costs -> 1 - c
    under ``cost_reverse``), ``returns`` / ``cost_returns`` = undiscounted to-go sums (gamma = 1)."""
    starts, lens = episode_segments(done_flags(dataset))
    out = []
    for s, n in zip(starts, lens):
        sl = slice(int(s), int(s + n))
        costs = np.asarray(dataset["costs"][sl], np.float32)
        if cost_reverse:
            costs = (1.0 - np.asarray(dataset["costs"][sl])).astype(np.float32)
        ep = dict(observations=np.asarray(dataset["observations"][sl], np.float32),
                  actions=np.asarray(dataset["actions"][sl], np.float32),
                  rewards=np.asarray(dataset["rewards"][sl], np.float32), costs=costs)
        ep["returns"] = discounted_cumsum(ep["rewards"], 1.0)
        ep["cost_returns"] = discounted_cumsum(ep["costs"], 1.0)
        out.append(ep)
    return out


def compute_cost_sample_prob(trajs: List[Dict[str, Array]], cost_transform=lambda x: 50 - x) -> Array:
    """dataset.py:439-459: p_i proportional to max(cost_transform(episode cost), 0)."""
    p = np.array([cost_transform(t["cost_returns"][0]) for t in trajs])
    p = np.where(p < 0, 0, p)
    return p / p.sum()


def compute_start_index_sample_prob(trajs: List[Dict[str, Array]], prob: float = 0.4) -> List[Array]:
    """dataset.py:472-494 (+ gauss_kernel :462-469): per trajectory, the start-index distribution of
    ``SequenceDataset(start_sampling=True)``: costs smoothed by exp(-j^2/10), |j| <= 10, plus an offset x that balances
    cost / no-cost steps for the target proportion ``prob``.  Pinned by tests/golden/samples.npz."""
    kern = np.exp(-(np.linspace(-10, 10, 21) ** 2 / 10.0))
    out = []
    for t in trajs:
        c = np.asarray(t["costs"])
        n, l = np.sum(c), len(c)
        x = 100 if prob * l - n <= 0 else n * (1 - prob) / (prob * l - n)
        if x <= 0:
            x = 1
        w = np.convolve(c, kern)[10:-10] + x
        out.append(w / w.sum())
    return out


BC_MODES = ("all", "multi-task", "safe", "risky", "boundary")


def process_bc_dataset(dataset: Dict[str, Array], cost_limit: float, gamma: float, bc_mode: str) -> Dict[str, Array]:
    """dataset.py:30-134 without the Pareto-frontier mode (needs oapackage): every transition of an episode gets the
    episode's discounted cost / reward return (:63-70; transitions after the last done keep 0), the mode selects
    transitions (:108-124), every array is filtered (:126-127) and multi-task appends the cost return to the
    observation (:128-130).  Returns a new dict (the reference edits its argument in place)."""
    if bc_mode not in BC_MODES:
        raise NotImplementedError(bc_mode)
    d = {k: np.asarray(v) for k, v in dataset.items()}
    starts, lens = episode_segments(done_flags(d))
    cr, rr = np.zeros_like(d["costs"]), np.zeros_like(d["rewards"])
    for s, n in zip(starts, lens):
        sl = slice(int(s), int(s + n))
        cr[sl] = discounted_cumsum(d["costs"][sl], gamma)[0]
        rr[sl] = discounted_cumsum(d["rewards"][sl], gamma)[0]
    d["cost_returns"], d["rew_returns"] = cr, rr
    if bc_mode in ("all", "multi-task"):
        keep = np.ones(cr.shape[0], bool)
    elif bc_mode == "safe":
        keep = cr <= cost_limit
    elif bc_mode == "risky":
        keep = cr >= 2 * cost_limit
    else:
        keep = np.logical_and(0.5 * cost_limit < cr, cr <= 1.5 * cost_limit)
    d = {k: v[keep] for k, v in d.items()}
    if bc_mode == "multi-task":
        d["observations"] = np.hstack((d["observations"], d["cost_returns"].reshape(-1, 1)))
    return d
