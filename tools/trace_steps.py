"""Kernel START times of an UN-PROFILED pipelined graph replay (csrc/trace.h under -DOSRL_TRACE):

    bash tools/build_trace_lib.sh
    OSRL_LIB=osrl_amd/lib/libosrl_trace.so python tools/trace_steps.py [config] [steps_per_graph] [replays]

Every instrumented launch's first workgroup leaves (kernel id | grid, 100 MHz real-time stamp, launch-site address) in a
device ring.  A launch site (the device-resident descriptor a captured launch reads) names one node of the graph, so the
records of R back-to-back replays align by (site, occurrence); printed: per node, the median start relative to the replay's
first record and the median distance to the NEXT start of a node of the same chain -- where a chain is a queue as the
rocprofv3 timeline of the same graph shows it (profiles/r6_timeline_5step_c2.txt): the order inside a chain is fixed.
Why: rocprofv3 intercepts the queues and rewrites packets; what a cross-queue wait costs differs with and without it.
"""
import ctypes as C
import os
import statistics
import sys
from collections import Counter, defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from osrl_amd import _lib as L

NAMES = {1: "step_begin", 2: "vae_ns_l0", 3: "vae_ns_fwd_enc", 40: "vae_ns_gen<0>", 41: "vae_ns_gen<1>", 42: "vae_ns_gen<2>",
         5: "mlp_fwd", 6: "mlp_fwd2", 7: "mlp_bwd_dz", 8: "mlp_fwd_nb", 9: "mlp_fwd_nb8", 11: "adam", 12: "cpq_ood_stat",
         13: "cpq_alpha_step", 14: "cpq_ood_select", 15: "cpq_ood_sum", 16: "polyak"}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    spg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dev = torch.device("cuda:0")
    lib = L.load()
    fn = getattr(lib, "osrl_debug_trace_set", None)
    if fn is None:
        raise SystemExit("this library has no trace support: OSRL_LIB=osrl_amd/lib/libosrl_trace.so (tools/build_trace_lib.sh)")
    fn.argtypes, fn.restype = [C.c_void_p, C.c_int64], C.c_int
    wl = bench.Workload(cfg, dev, 0, 1, None, n_store=1 << 16, use_graph=True, steps_per_graph=1)
    spg = wl.build_pipe(spg)
    cap = 1 << 16
    ring = torch.zeros(2 + 3 * cap, dtype=torch.int64, device=dev)
    wl.run(spg * 8)  # warm
    torch.cuda.synchronize()
    n_tu = fn(ring.data_ptr(), cap)
    assert n_tu > 0, n_tu
    torch.cuda.synchronize()
    t_host = []
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.run(spg * reps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = ring.cpu().numpy().astype(np.uint64)
    fn(None, 0)
    n = int(h[0])
    assert n <= cap, "ring too small"
    rec = h[2:2 + 3 * n].reshape(n, 3)
    ids = (rec[:, 0] & np.uint64(0xff)).astype(int)
    gx = ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xfffff)).astype(int)
    gy = ((rec[:, 0] >> np.uint64(28)) & np.uint64(0xfffff)).astype(int)
    t = rec[:, 1].astype(np.int64) * 0.01  # us
    site = rec[:, 2]
    order = np.argsort(t, kind="stable")
    ids, gx, gy, t, site = ids[order], gx[order], gy[order], t[order], site[order]
    per_rep = n // reps
    print(f"{cfg}: {spg} steps per graph, {reps} replays back to back, host clock {dt / (spg * reps) * 1e6:.1f} us per step; "
          f"{n} records = {per_rep} per replay ({n_tu} translation units armed)")
    # node = (site, k-th occurrence of that site inside a replay); replays are cut at the stamps' largest regular period:
    # the first record of a replay is the graph-opening step_begin -- the only step_begin whose site occurs... every spg-th
    # step_begin of engine 0; simpler and robust: cut by count (every replay issues the same per_rep launches)
    assert n == per_rep * reps, (n, per_rep, reps)
    # records of one replay are NOT contiguous in time order only if replays overlapped; they do not (a replay ends in a join)
    T = t.reshape(reps, per_rep)
    key = [(int(ids[i]), int(gx[i]), int(gy[i]), int(site[i])) for i in range(per_rep)]
    # align every replay to replay 0's node list by (site, occurrence)
    def nodes(lo):
        occ = Counter()
        out = []
        for i in range(lo, lo + per_rep):
            k = (int(ids[i]), int(gx[i]), int(gy[i]), int(site[i]))
            out.append((k, occ[k]))
            occ[k] += 1
        return out
    base = nodes(0)
    col = {nd: j for j, nd in enumerate(base)}
    M = np.full((reps, per_rep), np.nan)
    for r in range(reps):
        for i, nd in enumerate(nodes(r * per_rep)):
            M[r, col[nd]] = T[r, i] - T[r, 0]
    med = np.nanmedian(M[2:], axis=0)
    durs = np.diff(T[:, 0])
    print(f"replay start-to-start us: median {np.median(durs):.1f} = {np.median(durs) / spg:.1f} per step "
          f"(min {durs.min():.1f} max {durs.max():.1f})")
    rows = sorted(range(per_rep), key=lambda j: med[j])
    print(f"{'start us':>9s} {'+next':>7s}  kernel (grid)            site")
    for a, j in enumerate(rows):
        k, o = base[j]
        nxt = med[rows[a + 1]] - med[j] if a + 1 < len(rows) else float('nan')
        name = NAMES.get(k[0], f"mlp_dwt<{k[0] - 100}>" if k[0] > 100 else str(k[0]))
        print(f"{med[j]:9.1f} {nxt:7.1f}  {name:16s} {k[1]:4d} x {k[2]:<3d}   {k[3] & 0xffffff:06x}.{o}")


if __name__ == "__main__":
    main()
