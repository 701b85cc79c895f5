"""Per-replay duration of the first steps of a bench workload (HIP events between replays): how long does the step take
to reach its steady rate after the engine is built, after an idle gap, after a burst of other kernels?
    python tools/step_series.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
wl = bench.Workload(cfg, dev, 0, 1, None)


def series(n, label):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        wl.step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    print(f"{label}: first 12 us/step " + " ".join(f"{x:.0f}" for x in d[:12]) + f" | steps 13-25 mean {sum(d[12:25]) / 13:.0f}"
          f" | 26-60 mean {sum(d[25:60]) / 35:.0f} | last 100 mean {sum(d[-100:]) / 100:.0f}")


series(300, "right after the build (the first replay captures)")
series(300, "again, back to back")
time.sleep(0.5)
series(300, "after 0.5 s idle")
time.sleep(0.02)
series(300, "after 20 ms idle")
x = torch.randn(8192, 8192, device=dev)
for _ in range(40):
    y = x @ x
torch.cuda.synchronize()
series(300, "after 40 fp32 8k GEMMs + sync")
