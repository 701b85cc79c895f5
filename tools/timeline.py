"""Step timeline from a rocprofv3 --kernel-trace CSV: the kernels of one steady-state graph replay of the train step,
times relative to the step's first kernel.  Every column is the MEDIAN over all replays of the timed region that have
the same kernel sequence (a single replay is noisy: +-5 us per launch from one step to the next).
Usage: python tools/timeline.py <kernel_trace.csv>"""
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']),
       re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])[:30], r.get('Queue_Id', '?')) for r in rows]
ks.sort()
ticks = [i for i, k in enumerate(ks) if k[2].startswith('step_tick') or k[2].startswith('step_begin')]
steps = [ks[a:b] for a, b in zip(ticks[:-1], ticks[1:])]
# the most common kernel-name sequence = the replayed graph (the trace also holds warm-up / eager probes)
sig = lambda st: tuple(k[2] for k in st)  # noqa: E731
common = statistics.mode([sig(st) for st in steps])
steps = [st for st in steps if sig(st) == common]
nxt = {id(st): None for st in steps}
durs = []
for a, b in zip(ticks[:-1], ticks[1:]):
    if sig(ks[a:b]) == common:
        durs.append((ks[b][0] - ks[a][0]) / 1e3)
print(f"step duration us (median of {len(steps)} replays) {statistics.median(durs):.1f}  n kernels {len(common)}")
prev_end = {}
for j, name in enumerate(common):
    s = statistics.median((st[j][0] - st[0][0]) / 1e3 for st in steps)
    e = statistics.median((st[j][1] - st[0][0]) / 1e3 for st in steps)
    d = statistics.median((st[j][1] - st[j][0]) / 1e3 for st in steps)
    q = steps[0][j][3]
    gap = s - prev_end[q] if q in prev_end else 0.0
    prev_end[q] = e
    print(f"{s:7.1f} {e:7.1f} {d:6.1f} q{q} {name:30s} gap on its queue {gap:5.1f}")
