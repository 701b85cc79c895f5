"""Lab: api_path (trainer.train_one_step on device tensors) with / without a PipelinedSteps built on the engine before."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "pipe"
wl = bench.Workload("c2", dev, 0, 1, None, n_store=1 << 18, use_graph=True)
eng = wl.eng
if mode != "none":
    wl.build_pipe(5)
    if mode == "pipe_run":
        wl.run(40); torch.cuda.synchronize()
bench.preroll(dev, 40.0)
r = bench.api_path(wl)
print(mode.ljust(10), "api_path", r["steps_per_s"])
import cProfile, pstats
tr, batch = wl.trainer, wl.api_batch()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): tr.train_one_step(*batch)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io; s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print("\n".join(s.getvalue().splitlines()[4:26]))
