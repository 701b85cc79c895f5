#!/usr/bin/env python3
"""Kernel micro-benchmarks (HIP events) for tuning tile shapes / split counts on the GPU box.
Usage: python tools/kbench.py [--quick]   -> prints one line per (kernel, config)."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from osrl_amd import _lib as L  # noqa: E402
from osrl_amd.engine import glue as G  # noqa: E402
from osrl_amd.engine.core import DwPlan, FlatGroup, LayerRef, MlpRun, NetDesc, StepState  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def mk(E, dims, acts, dev, tile_rows=0):
    grp = FlatGroup("t", dev)
    for e in range(E):
        for l in range(len(dims) - 1):
            grp.add(f"{e}.{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{e}.{l}.w")
            grp.add(f"{e}.{l}.b", (dims[l + 1],))
    grp.finalize()
    grp.p.uniform_(-0.05, 0.05)
    grp.repack()
    nets = [[LayerRef(grp.view(f"{e}.{l}.w"), grp.view(f"{e}.{l}.b"), grp, f"{e}.{l}.w", f"{e}.{l}.b")
             for l in range(len(dims) - 1)] for e in range(E)]
    d = NetDesc(nets, acts, 1.0)
    d.c.tile_rows = tile_rows
    return grp, d


def glue_bench(dev):
    """The single-workgroup / row-local kernels on the CPQ step's latency chain at C2 sizes (B=2048, N=10, ad=2, L=4)
    plus the two N*B-row forwards and the 2048-row backward they wait for."""
    B, N, ad, Lz = 2048, 10, 2, 4
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    out, st = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
    x = r(N * B).abs()
    print(f"quantile n=20480: {timeit(lambda: G.quantile(x, N * B, 0.75, out)):.2f} us")
    qcs, kl = r(2, N * B), r(N * B).abs()
    print(f"cpq_ood_stat: {timeit(lambda: G.cpq_ood_stat(qcs, 2, kl, 0.75, N, B, B, out, out[1:])):.2f} us")
    head, eps, u, act = r(B, 2 * Lz), r(B, Lz), r(B, ad), r(B, ad)
    du, dh = torch.empty(B, ad, device=dev), torch.empty(B, 2 * Lz, device=dev)
    print(f"vae_loss: {timeit(lambda: G.vae_loss(u, act, head, B, ad, Lz, 0.5, B, du, st)):.2f} us")
    dzl = r(B, Lz)
    print(f"vae_latent_bwd: {timeit(lambda: G.vae_latent_bwd(head, eps, dzl, B, Lz, 0.5, B, dh)):.2f} us")
    hk, klr = r(N * B, 2 * Lz), torch.empty(N * B, device=dev)
    print(f"vae_kl_rows: {timeit(lambda: G.vae_kl_rows(hk, N * B, Lz, klr)):.2f} us")
    q2, qo, qco, rew, done = r(2, B), r(2, B), r(2, B), r(B), (torch.rand(B, device=dev) < 0.01).float()
    dq = torch.empty(2, B, device=dev)
    print(f"cpq_critic_loss: {timeit(lambda: G.cpq_critic_loss(qo, 2, qco, 2, q2, 2, rew, done, B, 0.99, 1.0, B, dq, st)):.2f} us")
    la = torch.zeros(1, device=dev)
    print(f"cpq_cost_loss: {timeit(lambda: G.cpq_cost_loss(qco, 2, q2, 2, None, rew, B, 0.99, 1.5, 1e-4, B, 1.0, la, dq, st)):.2f} us")
    print(f"cpq_actor_loss: {timeit(lambda: G.cpq_actor_loss(q2, 2, qco, 2, B, 1.0, B, dq, st)):.2f} us")
    hd, ea, tu, da = r(B, 2 * ad), r(B, ad), torch.tanh(r(B, ad)), r(2, B, ad)
    dhd = torch.empty(B, 2 * ad, device=dev)
    print(f"gauss_head_bwd: {timeit(lambda: G.gauss_head_bwd(hd, ea, tu, da, 2, B, ad, 1.0, dhd)):.2f} us")
    stt = StepState(dev, ["x"])
    stt.tick()
    for n, S in ((388812, 4), (172552, 8), (86532, 4)):  # vae / cost critics / actor groups of C2
        g = FlatGroup("a", dev, True)
        g.add("w", (n,))
        g.finalize()
        g.ensure_slabs(S)
        g.cur_splits = S
        print(f"adam n={n} S={S}: {timeit(lambda: g.adam_step(1e-3, stt.ptr, tau=0.005)):.2f} us")
    lin = lambda d: sum(a * b for a, b in zip(d[:-1], d[1:]))  # noqa: E731
    for name, E, dims, acts in (("enc", 1, [78, 400, 400, 8], ["relu", "relu", "id"]),
                                ("q x2", 2, [78, 256, 256, 1], ["relu", "relu", "id"])):
        grp, d = mk(E, dims, acts, dev, 80)
        x0, x1 = r(N * B, 76), r(N * B, 2)
        run = MlpRun(d, N * B, False, dev)
        t = timeit(lambda: run.forward(x0, x1))
        print(f"fwd {name} rows=20480 tile=80: {t:.2f} us  {2.0 * N * B * E * lin(dims) / t / 1e6:.2f} TF/s")
    for name, E, dims, acts in (("dec", 1, [80, 400, 400, 2], ["relu", "relu", "tanh"]),
                                ("q x2", 2, [78, 256, 256, 1], ["relu", "relu", "id"])):
        grp, d = mk(E, dims, acts, dev, 16)
        x0, x1 = r(B, 76), r(B, dims[0] - 76)
        run = MlpRun(d, B, True, dev)
        run.forward(x0, x1)
        run.setup_backward(r(E, B, dims[-1]), need_dz=True, dx_cols=(76, dims[0] - 76))
        print(f"fwd {name} rows=2048: {timeit(lambda: run.forward(x0, x1)):.2f} us | bwd_dz {timeit(run.backward_dz):.2f} us")


def main():
    dev = torch.device("cuda:0")
    if "--glue" in sys.argv:
        return glue_bench(dev)
    lin = lambda d: sum(a * b for a, b in zip(d[:-1], d[1:]))  # noqa: E731
    cfgs = [("q x2", 2, [78, 256, 256, 1], ["relu", "relu", "id"], 76),
            ("q x4", 4, [78, 256, 256, 1], ["relu", "relu", "id"], 76),
            ("actor", 1, [76, 256, 256, 4], ["relu", "relu", "id"], 76),
            ("enc", 1, [78, 400, 400, 8], ["relu", "relu", "id"], 76),
            ("dec", 1, [80, 400, 400, 2], ["relu", "relu", "tanh"], 76)]
    big_only = "--big" in sys.argv
    for name, E, dims, acts, d0 in cfgs:
        for rows in ((20480,) if big_only else (2048, 20480)):
            for tr in (16, 32, 64, 80):
                if tr == 64 and max(dims) > 256:
                    continue
                if tr == 80 and (rows == 2048 or max(dims) < 256):
                    continue
                grp, d = mk(E, dims, acts, dev, tr)
                x0 = torch.randn(rows, d0, device=dev)
                x1 = torch.randn(rows, dims[0] - d0, device=dev) if dims[0] > d0 else None
                run = MlpRun(d, rows, rows == 2048, dev)
                t_f = timeit(lambda: run.forward(x0, x1))
                fl = 2.0 * rows * E * lin(dims)
                line = f"fwd {name:6s} rows={rows:6d} tile={tr:2d}: {t_f:8.2f} us  {fl / t_f / 1e6:7.2f} TF/s"
                if rows == 2048:
                    dy = torch.randn(E, rows, dims[-1], device=dev)
                    run.setup_backward(dy, need_dz=True, dx_cols=(d0, dims[0] - d0) if x1 is not None else None)
                    t_b = timeit(run.backward_dz)
                    line += f" | bwd_dz {t_b:8.2f} us {2.0 * rows * E * (lin(dims) - dims[0] * dims[1]) / t_b / 1e6:7.2f} TF/s"
                    if tr == 16:
                        for ns in (1, 2, 4, 8, 16, 32):
                            plan = DwPlan(grp, run.dw_entries(), rows, dev, n_splits=ns)
                            t_w = timeit(plan.launch)
                            line += f" | dw[S={ns}] {t_w:6.1f}"
                print(line, flush=True)
    if big_only:
        return
    # quantile / adam
    x = torch.randn(20480, device=dev).abs()
    out = torch.zeros(4, device=dev)
    print(f"quantile n=20480: {timeit(lambda: G.quantile(x, 20480, 0.75, out)):.2f} us")
    x = torch.randn(163840, device=dev).abs()
    print(f"quantile n=163840: {timeit(lambda: G.quantile(x, 163840, 0.75, out)):.2f} us")
    from osrl_amd import _lib as L_
    ws = torch.zeros(L_.QUANTILE_WS, dtype=torch.int32, device=dev)
    print(f"quantile_ws (grid select) n=163840: {timeit(lambda: G.quantile_ws(x, 163840, 0.75, ws, out)):.2f} us")
    st = StepState(dev, ["x"])
    st.tick()
    for n, S in ((388812, 15), (388812, 4), (172552, 8), (86532, 1)):
        g = FlatGroup("a", dev, True)
        g.add("w", (n,))
        g.finalize()
        g.ensure_slabs(S)
        g.cur_splits = S
        print(f"adam n={n} S={S}: {timeit(lambda: g.adam_step(1e-3, st.ptr, tau=0.005)):.2f} us")


if __name__ == "__main__":
    main()
