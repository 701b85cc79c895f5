#!/usr/bin/env python3
"""Row-tile shape of the 2048-row multi-net forwards / backwards of the CPQ step: isolated time and workgroup-slot cost
(time x CUs occupied) per tile_rows (0 = the default 16-row 8-wave kernel).  python tools/tile_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import mk, timeit  # noqa: E402
from osrl_amd.engine.core import MlpRun  # noqa: E402

dev = torch.device("cuda:0")
rows = 2048
for E, dims, save in ((4, [78, 256, 256, 1], False), (6, [78, 256, 256, 1], False), (2, [78, 256, 256, 1], True),
                      (4, [78, 256, 256, 1], True), (1, [78, 400, 400, 8], True)):
    for tile in (0, 32, 64):
        grp, d = mk(E, dims, ["relu", "relu", "id"], dev, tile)
        x0, x1 = torch.randn(rows, 76, device=dev), torch.randn(rows, 2, device=dev)
        run = MlpRun(d, rows, save, dev)
        t_f = timeit(lambda: run.forward(x0, x1), 40)
        line = f"{E} nets {dims} rows {rows} save {int(save)} tile {tile or 16:2d}: fwd {t_f:6.2f} us"
        if save:
            dy = torch.randn(E, rows, dims[-1], device=dev)
            run.setup_backward(dy)
            t_b = timeit(run.backward_dz, 40)
            line += f"  bwd {t_b:6.2f} us"
        tr = tile or 16
        wgs = E * ((rows + tr - 1) // tr)
        line += f"  workgroups {wgs}"
        print(line, flush=True)
