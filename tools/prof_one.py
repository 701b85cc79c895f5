#!/usr/bin/env python3
"""Run N steps of one config for rocprofv3: prof_one.py {cdt|bcql|bearl|coptidice} [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.bench_all as ba  # noqa
from osrl_amd.algorithms import (BCQL, BEARL, CDT, BCQLTrainer, BEARLTrainer, CDTTrainer, COptiDICE,
                                 COptiDICETrainer)
DEV = "cuda:0"
which, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
rs = np.random.RandomState(0)
f = lambda *s: torch.tensor(rs.randn(*s), dtype=torch.float32, device=DEV)
if which == "cdt":
    B, T = 1024, 20
    m = CDT(11, 3, 1.0, seq_len=T, episode_len=1000, embedding_dim=256, num_layers=3, num_heads=8,
            attention_dropout=0.1, residual_dropout=0.1, embedding_dropout=0.1, use_rew=True,
            use_cost=True, cost_transform=True, stochastic=True, target_entropy=-3, device=DEV)
    tr = CDTTrainer(m, None, None, lr_warmup_steps=500, loss_cost_weight=0.02, stats_mode="none", use_graph=False)
    mask = torch.ones(B, T, device=DEV); mask[::10, T - 5:] = 0
    a = (f(B, T, 11), f(B, T, 3).clamp(-1, 1), torch.rand(B, T, device=DEV) * 10, torch.rand(B, T, device=DEV) * 20,
         torch.randint(0, 1000, (B, 1), device=DEV) + torch.arange(T, device=DEV)[None], mask, torch.rand(B, device=DEV),
         (torch.rand(B, T, device=DEV) < 0.1).float())
elif which == "bearl":  # train-config size (bearl_configs.py: batch 512, N = M = 10)
    B = 512
    m = BEARL(33, 8, 1.0, [256, 256], [256, 256], 400, 10, 0.99, 0.005, 0.5, 0.75, 50.0, 0.05, 10, [0.1, 0.003, 0.001],
              "gaussian", 2, 2, 10, 300, 0, device=DEV)
    tr = BEARLTrainer(m, None, None, 1e-3, 1e-3, 1e-3, 1e-3, stats_mode="none", use_graph=False)
    a = (f(B, 33), f(B, 33), f(B, 8).clamp(-1, 1), f(B), (torch.rand(B, device=DEV) < 0.1).float(), (torch.rand(B, device=DEV) < 0.01).float())
elif which == "coptidice":  # coptidice_configs.py: batch 512, 2 nu + 2 chi nets, softchi
    B = 512
    m = COptiDICE(33, 8, 1.0, "softchi", 0.01, np.ones((1, 33), np.float32), np.ones((1, 8), np.float32), [256, 256],
                  [256, 256], 0.99, 0.5, 0.01, 2, 2, 10, 300, device=DEV)
    tr = COptiDICETrainer(m, None, None, 1e-4, 1e-4, 1e-4, stats_mode="none", use_graph=False)
    a = ([f(B, 33), f(B, 33), f(B, 8).clamp(-1, 1), f(B), (torch.rand(B, device=DEV) < 0.1).float(),
          (torch.rand(B, device=DEV) < 0.01).float(), (torch.rand(B, device=DEV) < 0.01).float()],)
else:
    B = 4096
    m = BCQL(33, 8, 1.0, [256, 256], [256, 256], 400, 10, 0.99, 0.005, 0.05, 0.75, 0.5, [0.1, 0.003, 0.001], 2, 2, 10, 200, device=DEV)
    tr = BCQLTrainer(m, None, None, 1e-3, 1e-3, 1e-3, stats_mode="none", use_graph=False)
    a = (f(B, 33), f(B, 33), f(B, 8).clamp(-1, 1), f(B), (torch.rand(B, device=DEV) < 0.1).float(), (torch.rand(B, device=DEV) < 0.01).float())
for _ in range(n):
    tr.train_one_step(*a)
torch.cuda.synchronize()
