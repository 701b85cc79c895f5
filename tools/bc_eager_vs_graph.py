"""BC (8, 2) B=256: the one-launch step replayed as a hipGraph vs launched directly (one kernel per step either way)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osrl_amd.algorithms import BC
from osrl_amd.common.replay import ReplayStore, synthetic_transitions
dev = "cuda:0"
torch.manual_seed(0)
m = BC(8, 2, 1.0, [256, 256], 300, device=dev)
m.setup_optimizers(1e-3)
e = m.engine(256)
e.attach_replay(ReplayStore(synthetic_transitions(100000, 8, 2, seed=1), dev, seed=3))
for mode in ("graph", "eager", "graph", "eager"):
    g = mode == "graph"
    for _ in range(200):
        e.step_replay(use_graph=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3000
    for _ in range(n):
        e.step_replay(use_graph=g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode}: {dt / n * 1e6:.1f} us/step  {n / dt:.0f} steps/s")
