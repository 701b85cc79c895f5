"""Where the Trainer-API step spends its time over the replayed graph (bench.py `api_path`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from osrl_amd.common.logger import DummyLogger, store_stats


def timed(fn, n=400, w=40):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6


dev = torch.device("cuda", 0)
wl = Bn.Workload("c2", dev, 0, 1, None, n_store=1 << 18)
eng, tr = wl.eng, wl.trainer
print("replay step (gather in graph)  us/step, host us/step:", timed(wl.step))
eng.attach_replay(None)
eng.graph = None
batch = wl.api_batch()
eng.step(*batch)
print("graph only (static buffers)   :", timed(lambda: eng._run(True)))
print("load_batch + graph            :", timed(lambda: eng.step(*batch)))
tr.logger, tr.stats_mode = DummyLogger(), "lazy"
print("train_one_step lazy           :", timed(lambda: tr.train_one_step(*batch)))
tr.stats_mode = "none"
print("train_one_step none           :", timed(lambda: tr.train_one_step(*batch)))
print("host only: load_into          :", timed(lambda: eng.load_batch(*batch)))
