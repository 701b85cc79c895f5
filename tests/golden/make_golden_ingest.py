#!/usr/bin/env python3
"""Generate tests/golden/ingest.npz by IMPORTING the reference's dataset functions (never copying them):
process_sequence_dataset, compute_cost_sample_prob, process_bc_dataset (osrl/common/dataset.py) on the synthetic
dataset of tests/cases.py::make_ingest_dataset.  Build container only (needs /root/reference):

    python tests/golden/make_golden_ingest.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from cases import make_ingest_dataset  # noqa: E402
from make_golden import REF, _install_stubs  # noqa: E402

COST_LIMIT = 6.0


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    from osrl.common.dataset import compute_cost_sample_prob, process_bc_dataset, process_sequence_dataset
    out = {"meta": np.array([f"numpy {np.__version__}", f"torch {torch.__version__}", f"cost_limit {COST_LIMIT}"])}
    for rev in (False, True):
        data = make_ingest_dataset()
        traj, info = process_sequence_dataset(data, rev)
        tag = "rev" if rev else "fwd"
        out[f"seq_{tag}_len"] = np.array([len(t["costs"]) for t in traj], np.int64)
        for k in ("observations", "actions", "rewards", "costs", "returns", "cost_returns"):
            out[f"seq_{tag}_{k}"] = np.concatenate([t[k] for t in traj])
        out[f"seq_{tag}_prob50"] = np.asarray(compute_cost_sample_prob(traj, lambda x: 50 - x))
        out[f"seq_{tag}_prob8"] = np.asarray(compute_cost_sample_prob(traj, lambda x: 8 - x))      # clamps at 0
        out[f"seq_{tag}_probinv"] = np.asarray(compute_cost_sample_prob(traj, lambda x: 1 / (x + 10)))
    for mode in ("all", "multi-task", "safe", "risky", "boundary"):
        for gamma in (1.0, 0.99):
            data = make_ingest_dataset()
            data["index"] = np.arange(data["rewards"].shape[0])
            process_bc_dataset(data, COST_LIMIT, gamma, mode)
            tag = f"bc_{mode}_{gamma}"
            for k in ("index", "observations", "cost_returns", "rew_returns", "rewards"):
                out[f"{tag}_{k}"] = data[k]
    np.savez_compressed(os.path.join(HERE, "ingest.npz"), **out)
    print("wrote ingest.npz:", {k: v.shape for k, v in out.items() if k.startswith("seq_fwd") or "safe_1.0" in k})
    make_samples()


SEQ_LEN, RS, CS = 12, 0.1, 2.0


def sample_pairs(lens):
    """(trajectory, start) pairs: first / middle / last start of long, short (< seq_len) and 1-step trajectories --
    full windows, tail-padded windows, the n = 1 window."""
    order = np.argsort(lens, kind="stable")
    picks = list(order[:3]) + list(order[-3:]) + [int(order[len(order) // 2])]
    pairs = []
    for tr in picks:
        L_ = int(lens[tr])
        for st in sorted({0, L_ // 2, max(L_ - SEQ_LEN, 0), max(L_ - SEQ_LEN + 1, 0), max(L_ - 2, 0), L_ - 1}):
            pairs.append((int(tr), int(st)))
    return np.array(pairs, np.int64)


def make_samples():
    """tests/golden/samples.npz: outputs of the reference's OWN minibatch builders
    SequenceDataset._SequenceDataset__prepare_sample (dataset.py:749-775) and
    TransitionDataset._TransitionDataset__prepare_sample (dataset.py:832-842) on fixed (trajectory, start) / index
    lists, plus compute_start_index_sample_prob (dataset.py:472-494) -- pins oracle.prepare_sequence_sample,
    oracle.transition_sample and the device gathers."""
    from osrl.common.dataset import SequenceDataset, TransitionDataset, compute_start_index_sample_prob
    out = {"meta": np.array([f"numpy {np.__version__}", f"seq_len {SEQ_LEN}", f"reward_scale {RS}", f"cost_scale {CS}"])}
    for rev in (False, True):
        tag = "rev" if rev else "fwd"
        ds = SequenceDataset(make_ingest_dataset(), seq_len=SEQ_LEN, reward_scale=RS, cost_scale=CS, cost_reverse=rev)
        lens = np.array([len(t["costs"]) for t in ds.dataset])
        pairs = sample_pairs(lens)
        out[f"seq_{tag}_pairs"] = pairs
        names = ("states", "actions", "returns", "cost_returns", "time_steps", "mask", "episode_cost", "costs")
        cols = {n: [] for n in names}
        for tr, st in pairs:
            for n, v in zip(names, ds._SequenceDataset__prepare_sample(int(tr), int(st))):
                cols[n].append(np.asarray(v))
        for n in names:
            out[f"seq_{tag}_{n}"] = np.stack(cols[n])
        for prob in (0.4, 0.05):
            sp = compute_start_index_sample_prob(ds.dataset, prob)
            out[f"seq_{tag}_startprob_{prob}"] = np.concatenate([np.asarray(p, np.float64) for p in sp])
    for init in (False, True):
        data = make_ingest_dataset()
        td = TransitionDataset(data, reward_scale=RS, cost_scale=CS, state_init=init)
        n = td.dataset_size
        idx = np.array(sorted({0, 1, 2, n // 3, n // 2, n - 2, n - 1} | set(np.flatnonzero(td.dataset["done"])[:6].tolist())
                              | set((np.flatnonzero(td.dataset["done"])[:6] + 1).clip(0, n - 1).tolist())), np.int64)
        tag = "init" if init else "plain"
        out[f"trans_{tag}_idx"] = idx
        names = ("observations", "next_observations", "actions", "rewards", "costs", "done") + (("is_init",) if init else ())
        cols = {k: [] for k in names}
        for i in idx:
            for k, v in zip(names, td._TransitionDataset__prepare_sample(int(i))):
                cols[k].append(np.asarray(v))
        for k in names:
            out[f"trans_{tag}_{k}"] = np.stack(cols[k])
        if init:
            p, os_, as_ = td.get_dataset_states()
            out["trans_init_states"] = np.concatenate([[p], np.ravel(os_), np.ravel(as_)]).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "samples.npz"), **out)
    print("wrote samples.npz:", {k: v.shape for k, v in out.items() if "fwd" in k or "init" in k})


if __name__ == "__main__":
    main()
