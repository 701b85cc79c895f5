#!/usr/bin/env python3
"""Tail-quantisation probe: TF/s of one fused-MLP forward config as a function of the row count
(= number of workgroups) at a fixed tile size.  usage: ktail.py NAME TILE ROWS..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import mk, timeit  # noqa: E402
from osrl_amd.engine.core import MlpRun  # noqa: E402

CFG = {"q2": (2, [78, 256, 256, 1], ["relu", "relu", "id"], 76), "q4": (4, [78, 256, 256, 1], ["relu", "relu", "id"], 76),
       "enc": (1, [78, 400, 400, 8], ["relu", "relu", "id"], 76), "actor": (1, [76, 256, 256, 4], ["relu", "relu", "id"], 76)}
name, tile = sys.argv[1], int(sys.argv[2])
E, dims, acts, d0 = CFG[name]
dev = torch.device("cuda:0")
lin = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
for rows in map(int, sys.argv[3:]):
    grp, d = mk(E, dims, acts, dev, tile)  # 0 = library default, -1 = opt-in LDS-staged kernel
    x0 = torch.randn(rows, d0, device=dev)
    x1 = torch.randn(rows, dims[0] - d0, device=dev) if dims[0] > d0 else None
    run = MlpRun(d, rows, False, dev)
    t = timeit(lambda: run.forward(x0, x1))
    print(f"{name} tile={tile} rows={rows:6d} wgs={(rows + max(tile, 1) - 1) // max(tile, 1) * E:5d}: {t:8.2f} us {2.0 * rows * E * lin / t / 1e6:7.2f} TF/s",
          flush=True)
