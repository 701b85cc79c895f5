"""Isolated durations of the N*B-row launches on plain tiles and on tiles of shared observations (osrl_rows_t.share0):
OSRL_LAB=1 OSRL_OOD_SHARE=1 python tools/prefix_bench.py [config]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from osrl_amd import _lib as L

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
wl = bench.Workload(cfg, dev, 0, 1, None, n_store=1 << 16, use_graph=True, steps_per_graph=1)
e = wl.eng
wl.run(3)
torch.cuda.synchronize()
B = e.B


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, run, on in (("cost_old x2 (256-wide)", e.r_costold_ood, e.pre_cost), ("vae encoder (400-wide)", e.r_enc_ood, e.pre_enc)):
    t0 = timeit(lambda: run.forward(e.obs, e.sampled, map0=L.MAP_MOD, div0=B))
    print(f"{name}: plain launch {t0:.1f} us", end="")
    if on:
        t1 = timeit(lambda: run.forward(e.obs, e.sampled, map0=L.MAP_MOD, div0=B, share_k16=on))
        print(f"; on tiles of shared observations ({on} of layer 0's k-steps once per observation) {t1:.1f} us")
    else:
        print()
