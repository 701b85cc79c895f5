"""Timeline of a MULTI-step graph replay (engine/pipeline.py) from a rocprofv3 --kernel-trace CSV.

    python tools/timeline_graph.py <kernel_trace.csv> <steps_per_graph>

A replay = the kernels from one graph-opening prologue (every steps_per_graph-th step_begin launch, the one that follows the
longest idle gap) to the next.  Kernels are ordered PER QUEUE (the order inside a queue is fixed by the graph; the interleaving
of the two queues jitters from replay to replay); every column is the median over all replays with the most common per-queue
kernel sequence, times relative to the replay's first kernel.  Also printed: per-queue busy time and the stretches where
only ONE queue has a kernel running."""
import csv
import re
import statistics
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
spg = int(sys.argv[2])
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']),
       re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])[:34], r.get('Queue_Id', '?')) for r in rows]
ks.sort()
begins = [i for i, k in enumerate(ks) if k[2].startswith('step_begin')]
# idle gap in front of each prologue launch: the graph-opening one has the longest (nothing of the previous replay runs)
def idle_before(i):
    s = ks[i][0]
    last_end = max((k[1] for k in ks[max(0, i - 40):i]), default=s)
    return s - last_end
# choose the phase (mod spg) whose begins have the largest median idle gap
best = max(range(spg), key=lambda ph: statistics.median([idle_before(i) for i in begins[ph::spg]] or [0]))
opens = begins[best::spg]
reps = [ks[a:b] for a, b in zip(opens[:-1], opens[1:])]
def per_queue(rep):
    d = defaultdict(list)
    for k in rep:
        d[k[3]].append(k)
    return d
sig = lambda rep: tuple(sorted((q, tuple(k[2] for k in v)) for q, v in per_queue(rep).items()))  # noqa: E731
common = Counter(sig(r) for r in reps).most_common(1)[0][0]
reps = [r for r in reps if sig(r) == common]
durs = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(reps[:-1], reps[1:])]
n_k = sum(len(v) for _, v in common)
print(f"replay duration us (median of {len(reps)} replays of {spg} steps) {statistics.median(durs):.1f} = "
      f"{statistics.median(durs) / spg:.1f} per step; n kernels {n_k}")
ds = sorted(durs)
if ds:
    print("replay start-to-start us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f mean %.1f (n %d)" % (
        ds[0], ds[len(ds) // 10], ds[len(ds) // 2], ds[(9 * len(ds)) // 10], ds[-1], sum(ds) / len(ds), len(ds)))
ent = []
for q, names in common:
    pq = [per_queue(r)[q] for r in reps]
    for j, name in enumerate(names):
        s = statistics.median((r[j][0] - rep[0][0]) / 1e3 for r, rep in zip(pq, reps))
        e = statistics.median((r[j][1] - rep[0][0]) / 1e3 for r, rep in zip(pq, reps))
        ent.append((s, e, q, name))
ent.sort()
prev_end = {}
for s, e, q, name in ent:
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f} q{q} {name:34s} gap on its queue {gap:6.1f}")
# overlap accounting on the median timeline
ev = sorted([(s, 1, q) for s, e, q, _ in ent] + [(e, -1, q) for s, e, q, _ in ent])
active, t_prev, alone, both, idle = Counter(), ent[0][0], Counter(), 0.0, 0.0
for t, d, q in ev:
    live = [x for x, c in active.items() if c > 0]
    if len(live) == 1:
        alone[live[0]] += t - t_prev
    elif len(live) >= 2:
        both += t - t_prev
    else:
        idle += t - t_prev
    active[q] += d
    t_prev = t
print("only one queue running: " + ", ".join(f"q{q} {v:.1f} us" for q, v in sorted(alone.items())) +
      f"; two queues running {both:.1f} us; none {idle:.1f} us")
