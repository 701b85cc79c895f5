import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osrl_amd.algorithms import CPQ
torch.manual_seed(0)
m = CPQ(76, 2, 1.0, [256, 256], [256, 256], 400, 10, episode_len=200, device="cuda:0")
fp = m.fast_policy()
obs = np.random.randn(76).astype(np.float32)
acc = np.zeros(3)
for i in range(2200):
    fp.act(obs)
    if i >= 200:
        acc += fp.logp_out[1:4]
print("us: stage-in %.2f  net %.2f  head+fence %.2f" % tuple(acc / 2000))
