"""How much does the replayed two-branch step depend on WHICH streams / hardware queues the capture lands on?

The driver's round-2 box measured 1700 steps/s for the c2 step, the build boxes 2155, and only the graphs with a side
branch differed.  Hypothesis: the graph's branches are executed on HIP streams that share a small pool of hardware
queues (GPU_MAX_HW_QUEUES); whether the two branches land on DIFFERENT queues depends on how many streams the process
created before the capture (bench.py's eager in-step probe creates one).  This script re-captures the same step
several times in one process, with k extra streams created (and used once) in between, and times each capture.

    python tools/graph_variance.py [config] [n_captures]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def rate(eng, n=150):
    for _ in range(10):
        eng.step_replay(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step_replay(True)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n_cap = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.Workload(name, dev, 0, 1, None, n_store=1 << 18)
    eng = wl.eng
    keep = []
    print(f"env GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} config={name}", flush=True)
    for i in range(n_cap):
        eng.graph = None
        r = rate(eng)
        print(f"capture {i}: {r:8.1f} steps/s   (extra streams created so far: {len(keep)})", flush=True)
        # one more stream, used once, before the next capture
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(16, device=dev).add_(1.0)
        torch.cuda.synchronize()
        keep.append(s)
    if name in ("c2", "c4"):
        # what bench.py does in front of its timed region
        mean_us, med_us = bench.in_step_us(eng)
        eng.graph = None
        print(f"after the eager in-step probe ({mean_us:.1f} us): {rate(eng):8.1f} steps/s", flush=True)
        eng.parallel_branches = False
        eng.graph = None
        print(f"single chain (no side branch): {rate(eng):8.1f} steps/s", flush=True)


if __name__ == "__main__":
    main()
