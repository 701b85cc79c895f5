"""One-off dataset ingestion timings: device kernels (csrc/ingest.hip) vs the numpy oracle on the host.
1 M transitions at the C2 dimensions (obs 76, act 2), ~1000-step episodes.

    python tests/bench_ingest.py > profiles/rN_ingest_bench.json

(Kept under tests/: its CPU leg runs the oracle, which only tests/, smoke() and bench.py's cpu_baseline may use.)
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from oracle import ingest_oracle as IO
    from osrl_amd.common.ingest import compute_cost_sample_prob, process_bc_dataset, process_sequence_dataset
    n, od, ad = 1_000_000, 76, 2
    rs = np.random.RandomState(0)
    f = np.float32
    data = dict(observations=rs.randn(n, od).astype(f), next_observations=rs.randn(n, od).astype(f),
                actions=rs.uniform(-1, 1, (n, ad)).astype(f), rewards=rs.uniform(0, 1, n).astype(f),
                costs=(rs.uniform(size=n) < 0.05).astype(f), terminals=np.zeros(n, f),
                timeouts=(np.arange(n) % 1000 == 999).astype(f))
    dev = {k: torch.from_numpy(v).cuda() for k, v in data.items()}
    torch.cuda.synchronize()

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    out = {"transitions": n, "obs_dim": od, "rows": []}
    t = timed(lambda: compute_cost_sample_prob(process_sequence_dataset(dev, False, "cuda:0"), ("affine", -1.0, 70.0)))
    out["rows"].append(dict(what="process_sequence_dataset + compute_cost_sample_prob", path="device (arrays resident)",
                            seconds=round(t, 5)))
    t = timed(lambda: process_bc_dataset(dev, 40.0, 0.99, "safe", "cuda:0"))
    out["rows"].append(dict(what='process_bc_dataset(mode="safe")', path="device (arrays resident)", seconds=round(t, 5),
                            note="includes the stable compaction and the gather of every field (~0.6 GB moved)"))
    t0 = time.perf_counter()
    tr = IO.process_sequence_dataset(data, False)
    IO.compute_cost_sample_prob(tr, lambda x: 70 - x)
    out["rows"].append(dict(what="process_sequence_dataset + compute_cost_sample_prob", path="numpy oracle, 1 core",
                            seconds=round(time.perf_counter() - t0, 4)))
    t0 = time.perf_counter()
    IO.process_bc_dataset(data, 40.0, 0.99, "safe")
    out["rows"].append(dict(what='process_bc_dataset(mode="safe")', path="numpy oracle, 1 core",
                            seconds=round(time.perf_counter() - t0, 4)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
