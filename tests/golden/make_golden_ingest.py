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


if __name__ == "__main__":
    main()
