#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, mean/min/max duration, and how many of
the dispatches ran alone on the GPU (no other kernel overlapping) -- bench.py's roofline launches are the
isolated ones; the same kernels inside the replayed step graph overlap with the other graph branch and are slower.

    python tools/trace_summary.py gpurun_out/prof/x_kernel_trace.csv > profiles/rN_bench_trace_summary.txt
"""
import csv
import re
import sys
from collections import defaultdict


def main(path):
    rows = list(csv.DictReader(open(path)))
    ks = []
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)
        grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid))
    ks.sort()
    # overlap test against neighbours in start order
    alone = [True] * len(ks)
    max_end = -1
    max_i = -1
    for i, (s, e, _, _) in enumerate(ks):
        if s < max_end:
            alone[i] = False
            alone[max_i] = False
        if e > max_end:
            max_end, max_i = e, i
    agg = defaultdict(lambda: [[], []])
    for (s, e, n, g), a in zip(ks, alone):
        agg[(n, g)][0 if a else 1].append((e - s) / 1e3)
    print(f"{'kernel':34s} {'grid(wg x nets)':>16s} | {'isolated: n':>11s} {'mean us':>8s} {'min':>7s} {'max':>7s} | "
          f"{'overlapped: n':>13s} {'mean us':>8s}")
    tot = lambda v: sum(v[0]) + sum(v[1])  # noqa: E731
    for (n, g), v in sorted(agg.items(), key=lambda kv: -tot(kv[1])):
        iso, ov = v
        f = lambda x: (f"{len(x):11d} {sum(x) / len(x):8.2f} {min(x):7.2f} {max(x):7.2f}" if x else  # noqa: E731
                       f"{0:11d} {'-':>8s} {'-':>7s} {'-':>7s}")
        o = f"{len(ov):13d} {sum(ov) / len(ov):8.2f}" if ov else f"{0:13d} {'-':>8s}"
        print(f"{n[:34]:34s} {str(g[0]) + ' x ' + str(g[1]):>16s} | {f(iso)} | {o}")


if __name__ == "__main__":
    main(sys.argv[1])
