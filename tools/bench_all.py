#!/usr/bin/env python3
"""grad-steps/s of every BASELINE.json config that fits one GPU (C1 BC, C2 CPQ, C3 BCQ-Lag, C5 CDT), graph mode,
synthetic batches resident in HBM.  Prints one line per config (not the driver's bench contract -- see bench.py)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osrl_amd.algorithms import (BC, BCQL, BEARL, CDT, CPQ, BCQLTrainer, BCTrainer, BEARLTrainer, CDTTrainer,  # noqa: E402
                                 COptiDICE, COptiDICETrainer, CPQTrainer)

DEV = "cuda:0"


def run(name, step, n=100, warm=10):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {1 / dt:9.1f} grad-steps/s   {dt * 1e3:8.3f} ms/step", flush=True)


def main():
    rs = np.random.RandomState(0)
    f = lambda *s: torch.tensor(rs.randn(*s), dtype=torch.float32, device=DEV)  # noqa: E731
    torch.manual_seed(0)
    # C1 BC (8,2) B=256
    m = BC(8, 2, 1.0, [256, 256], 300, device=DEV)
    tr = BCTrainer(m, None, None, actor_lr=1e-3, stats_mode="none")
    o, a = f(256, 8), f(256, 2).clamp(-1, 1)
    run("C1 BC   (8,2)   B=256 ", lambda: tr.train_one_step(o, a), 500, 20)
    # C2 CPQ (76,2) B=2048
    m = CPQ(76, 2, 1.0, [256, 256], [256, 256], 400, 10, 0.99, 0.005, 0.5, 2, 2, 1.5, 10, 1000, device=DEV)
    tr = CPQTrainer(m, None, None, 1e-4, 1e-3, 1e-4, 1e-3, stats_mode="none")
    B = 2048
    args = (f(B, 76), f(B, 76), f(B, 2).clamp(-1, 1), f(B), (torch.rand(B, device=DEV) < 0.1).float(),
            (torch.rand(B, device=DEV) < 0.01).float())
    run("C2 CPQ  (76,2)  B=2048", lambda: tr.train_one_step(*args), 300, 20)
    # C3 BCQ-Lag (33,8) B=4096
    m = BCQL(33, 8, 1.0, [256, 256], [256, 256], 400, 10, 0.99, 0.005, 0.05, 0.75, 0.5, [0.1, 0.003, 0.001], 2, 2, 10,
             200, device=DEV)
    tr = BCQLTrainer(m, None, None, 1e-3, 1e-3, 1e-3, stats_mode="none")
    B = 4096
    args3 = (f(B, 33), f(B, 33), f(B, 8).clamp(-1, 1), f(B), (torch.rand(B, device=DEV) < 0.1).float(),
             (torch.rand(B, device=DEV) < 0.01).float())
    run("C3 BCQL (33,8)  B=4096", lambda: tr.train_one_step(*args3), 100, 10)
    # BEAR-Lag at its train-config defaults (bearl_configs.py: B=512, N=M=10, hidden 256, vae 400) on the C3 dims
    m = BEARL(33, 8, 1.0, [256, 256], [256, 256], 400, 10, 0.99, 0.005, 0.5, 0.75, 50.0, 0.05, 10, [0.1, 0.003, 0.001],
              "gaussian", 2, 2, 10, 300, 0, device=DEV)
    tr = BEARLTrainer(m, None, None, 1e-3, 1e-3, 1e-3, 1e-3, stats_mode="none")
    for B in (512, 4096):
        argsb = tuple(t[:B].contiguous() for t in args3)
        run(f"BEAR-L  (33,8)  B={B:<4d}", lambda: tr.train_one_step(*argsb), 100, 10)
    # COptiDICE at its train-config defaults (coptidice_configs.py: B=512, hidden 256, 2 nu + 2 chi nets, softchi)
    m = COptiDICE(33, 8, 1.0, "softchi", 0.01, np.ones((1, 33), np.float32), np.ones((1, 8), np.float32), [256, 256],
                  [256, 256], 0.99, 0.5, 0.01, 2, 2, 10, 300, device=DEV)
    tr = COptiDICETrainer(m, None, None, 1e-4, 1e-4, 1e-4, stats_mode="none")
    B = 512
    batch = [t[:B].contiguous() for t in args3] + [(torch.rand(B, device=DEV) < 0.01).float()]
    run("COptiDICE (33,8) B=512", lambda: tr.train_one_step(batch), 300, 20)
    # C5 CDT (11,3) T=20 E=256 8 heads 3 layers B=1024, dropout 0.1 (cdt_configs.py:28-30; pass a 3rd argv to override)
    B, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 20
    pdrop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    m = CDT(11, 3, 1.0, seq_len=T, episode_len=1000, embedding_dim=256, num_layers=3, num_heads=8,
            attention_dropout=pdrop, residual_dropout=pdrop, embedding_dropout=pdrop, use_rew=True,
            use_cost=True, cost_transform=True, stochastic=True, target_entropy=-3, device=DEV)
    tr = CDTTrainer(m, None, None, learning_rate=1e-4, weight_decay=1e-4, clip_grad=0.25, lr_warmup_steps=500,
                    loss_cost_weight=0.02, stats_mode="none")
    start = torch.randint(0, 1000, (B, 1), device=DEV)
    mask = torch.ones(B, T, device=DEV)
    mask[::10, T - 5:] = 0
    a5 = (f(B, T, 11), f(B, T, 3).clamp(-1, 1), torch.rand(B, T, device=DEV) * 10, torch.rand(B, T, device=DEV) * 20,
          start + torch.arange(T, device=DEV)[None], mask, torch.rand(B, device=DEV) * 20,
          (torch.rand(B, T, device=DEV) < 0.1).float())
    run(f"C5 CDT  (11,3)  B={B} T=20 E=256", lambda: tr.train_one_step(*a5), 10, 2)
    print("CDT last stats:", m._engine.st.read_stats())


if __name__ == "__main__":
    main()
