"""Isolated timings of the CPQ VAE group's weight-gradient launch under several plans (64 x 64 tiles x common split count
vs the flat work list on 80 x 80 / 64 x 64 tiles).   python tools/dw_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from osrl_amd.engine.core import DwPlan  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    wl = bench.Workload("c2", dev, 0, 1, None, n_store=1 << 18)
    eng = wl.eng
    for _ in range(3):
        eng.step_replay(False)
    torch.cuda.synchronize()
    g = eng.model.groups["vae"]
    ents = eng.r_enc.dw_entries() + eng.r_dec.dw_entries()
    B = eng.B
    flops = 2.0 * B * sum(g.layout[e[2]][1][0] * g.layout[e[2]][1][1] for e in ents)
    plans = [("64x64 tiles, 2 splits (round 2)", dict(n_splits=2)), ("64x64 tiles, 3 splits", dict(n_splits=3)),
             ("64x64 tiles, 4 splits", dict(n_splits=4))]
    for T in (5, 4):
        for s in (2, 3, 4, 6, 8):
            plans.append((f"flat list, {16*T}x{16*T} tiles, full tile = {s} splits", dict(n_splits=s, tile_blocks=T)))
    for name, kw in plans:
        p = DwPlan(g, ents, B, dev, **kw)
        t = bench.time_kernel(p.launch, iters=50)
        n = p.n_work if p.n_work else p.n_items * p.n_splits_small
        print(f"{name:52s} {n:4d} workgroups, {p.n_splits} slabs: {t*1e6:7.2f} us  {flops/t/1e12:6.1f} TF/s")


if __name__ == "__main__":
    main()
