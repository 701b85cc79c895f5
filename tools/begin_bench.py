"""Step prologue: the three separate launches vs osrl_step_begin (tools/kbench.py timing helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import timeit
from osrl_amd.common.replay import ReplayStore, synthetic_transitions
from osrl_amd.engine.core import StepState, randn_fill

dev = torch.device("cuda", 0)
B, od, ad = 2048, 76, 2
store = ReplayStore(synthetic_transitions(1 << 18, od, ad), dev)
dst = [torch.zeros(B, w, device=dev) for w in (od, od, ad, 1, 1, 1)]
noise = torch.zeros(B * 30, device=dev)
st = StepState(dev, ["a", "b", "c", "d", "e"])
print(f"tick                 : {timeit(st.tick):.2f} us")
print(f"randn                : {timeit(lambda: randn_fill(noise, 1, 0, st.ptr)):.2f} us")
print(f"gather               : {timeit(lambda: store.gather(dst, st.ptr)):.2f} us")
ga = store.gather_args(dst)
print(f"begin(tick only)     : {timeit(lambda: st.begin()):.2f} us")
print(f"begin(noise)         : {timeit(lambda: st.begin(noise, 1, 0)):.2f} us")
print(f"begin(gather)        : {timeit(lambda: st.begin(None, 0, 0, ga)):.2f} us")
print(f"begin(noise+gather)  : {timeit(lambda: st.begin(noise, 1, 0, ga)):.2f} us")
