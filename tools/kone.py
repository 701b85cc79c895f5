#!/usr/bin/env python3
"""Run ONE fused-MLP forward config repeatedly (for rocprofv3 --pmc passes).
usage: kone.py NAME ROWS TILE [ITERS]   NAME in {q2,q4,enc,actor}"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import mk  # noqa: E402
from osrl_amd.engine.core import MlpRun  # noqa: E402

CFG = {"q2": (2, [78, 256, 256, 1], ["relu", "relu", "id"], 76), "q4": (4, [78, 256, 256, 1], ["relu", "relu", "id"], 76),
       "enc": (1, [78, 400, 400, 8], ["relu", "relu", "id"], 76), "actor": (1, [76, 256, 256, 4], ["relu", "relu", "id"], 76)}
name, rows, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
E, dims, acts, d0 = CFG[name]
dev = torch.device("cuda:0")
grp, d = mk(E, dims, acts, dev, tile)
x0 = torch.randn(rows, d0, device=dev)
x1 = torch.randn(rows, dims[0] - d0, device=dev) if dims[0] > d0 else None
run = MlpRun(d, rows, False, dev)
for _ in range(iters):
    run.forward(x0, x1)
torch.cuda.synchronize()
