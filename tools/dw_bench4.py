"""Isolated timings of the four CPQ groups' dW launches as the step configures them, plus alternatives (round 4:
the fragment ring of dwt_tile).   python tools/dw_bench4.py   (OSRL_LIB selects a library build)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from osrl_amd.engine.core import DwPlan  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    wl = bench.Workload(os.environ.get("CFG", "c2"), dev, 0, 1, None, n_store=1 << 18)
    eng = wl.eng
    for _ in range(3):
        eng.step_replay(False)
    torch.cuda.synchronize()
    B = eng.B
    groups = {"vae": eng.r_enc.dw_entries() + eng.r_dec.dw_entries(), "critic": eng.r_critic.dw_entries(),
              "cost_critic": eng.r_cost.dw_entries(), "actor": eng.r_actor_obs.dw_entries()}
    for gname, ents in groups.items():
        g = eng.model.groups[gname]
        flops = 2.0 * B * sum(g.layout[e[2]][1][0] * g.layout[e[2]][1][1] for e in ents)
        variants = [(5, 3), (4, 4), (4, 8), (3, 4), (2, 2), (2, 4)] if gname == "vae" else \
                   [(4, 8), (4, 4), (3, 4), (2, 2), (2, 4), (2, 1)]
        for T, s in variants:
            p = DwPlan(g, ents, B, dev, n_splits=s, tile_blocks=T)
            t = bench.time_kernel(p.launch, iters=40)
            print(f"{gname:12s} T={T} splits={s:2d} workgroups {p.n_work:4d}: {t * 1e6:7.2f} us  "
                  f"{flops / t / 1e12:6.1f} TF/s = {flops / t / 1e12 / bench.PEAK_FP32_TFLOPS:.3f}", flush=True)


if __name__ == "__main__":
    main()
