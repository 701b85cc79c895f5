#!/usr/bin/env python3
"""Latency of the B=1 act() path: C call alone, FastPolicy.act(), model.act(), full episode loop (GPU box)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osrl_amd.algorithms import BC, BCQL, CPQ, CPQTrainer
from osrl_amd.common.synthetic_env import SyntheticSafeEnv
from osrl_amd.engine.core import cur_stream

dev = "cuda:0"
torch.manual_seed(0)
m = CPQ(76, 2, 1.0, [256, 256], [256, 256], 400, 10, episode_len=200, device=dev)
fp = m.fast_policy()
obs = np.random.randn(76).astype(np.float32)
N = 20000
for _ in range(100):
    fp.act(obs)
st = cur_stream()
t0 = time.perf_counter()
for _ in range(N):
    fp._fn(fp._h, 1, 1, 0, 0, st)
print(f"C call (launch + kernel + spin): {(time.perf_counter() - t0) / N * 1e6:.2f} us")
t0 = time.perf_counter()
for _ in range(N):
    fp.act(obs)
print(f"FastPolicy.act: {(time.perf_counter() - t0) / N * 1e6:.2f} us")
t0 = time.perf_counter()
for _ in range(N):
    m.act(obs, True, True)
print(f"CPQ.act: {(time.perf_counter() - t0) / N * 1e6:.2f} us")
t0 = time.perf_counter()
for _ in range(N):
    cur_stream()
print(f"cur_stream(): {(time.perf_counter() - t0) / N * 1e6:.2f} us")
env = SyntheticSafeEnv(76, 2, 200, seed=1, init_noise=0.5)
o, _ = env.reset()
a = np.zeros(2, np.float32)
t0 = time.perf_counter()
for _ in range(N):
    o, r, te, tr_, info = env.step(a)
    if te or tr_:
        o, _ = env.reset()
print(f"env.step (numpy): {(time.perf_counter() - t0) / N * 1e6:.2f} us")
tr = CPQTrainer(m, env, None, device=dev)
t0 = time.perf_counter()
tr.evaluate(50)
dt = time.perf_counter() - t0
print(f"episode loop: {50 * 200 / dt:.0f} env-steps/s")
b = BCQL(33, 8, 1.0, [256, 256], [256, 256], 400, 10, episode_len=200, device=dev)
ob = np.random.randn(33).astype(np.float32)
b.act(ob)
t0 = time.perf_counter()
for _ in range(N):
    b.act(ob)
print(f"BCQL.act: {(time.perf_counter() - t0) / N * 1e6:.2f} us")
