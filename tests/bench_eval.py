"""Evaluation throughput (env-steps/s) of the batched, graph-captured evaluate() against the reference-shaped
episode-by-episode loop (one B=1 policy call + host round trip per env step) and against the CPU oracle policy
driving the numpy environment.  CPQ at the C2 dimensions (obs 76, act 2, hidden [256,256]).

    python tests/bench_eval.py [--episodes 1024] [--len 200] > profiles/rN_eval_bench.json

(Kept under tests/: its CPU leg runs the oracle, which only tests/, smoke() and bench.py's cpu_baseline may use.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=1024)
    ap.add_argument("--len", type=int, default=200)
    ap.add_argument("--scalar-episodes", type=int, default=20)
    a = ap.parse_args()
    from osrl_amd.algorithms import BCQL, CPQ, BCQLTrainer, CPQTrainer
    from osrl_amd.common.logger import DummyLogger
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    dev = "cuda:0"
    out = {"episode_len": a.len, "rows": []}
    for algo, od, ad in (("cpq", 76, 2), ("bcql", 33, 8)):
        torch.manual_seed(0)
        if algo == "cpq":
            m = CPQ(od, ad, 1.0, [256, 256], [256, 256], 400, 10, episode_len=a.len, device=dev)
            tr = CPQTrainer(m, None, DummyLogger(), device=dev)
        else:
            m = BCQL(od, ad, 1.0, [256, 256], [256, 256], 400, 10, episode_len=a.len, device=dev)
            tr = BCQLTrainer(m, None, DummyLogger(), device=dev)
        env = SyntheticSafeEnv(od, ad, a.len, seed=1, init_noise=0.5)
        for E in sorted({1, 64, a.episodes}):
            tr.env = VecSyntheticSafeEnv(env, E, dev)
            tr.evaluate(E)  # builds + captures
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                r = tr.evaluate(E)
            dt = (time.perf_counter() - t0) / reps
            out["rows"].append(dict(algo=algo, path="batched-graph", episodes=E, s_per_eval=round(dt, 5),
                                    env_steps_per_s=round(E * a.len / dt, 1), us_per_env_step_launch=round(dt / a.len * 1e6, 2),
                                    mean_return=round(r[0], 4), mean_cost=round(r[1], 4)))
        tr.env = env  # reference-shaped scalar loop on the same HIP ops
        tr.evaluate(1)  # builds the resident policy descriptor (pinned I/O block): steady state is what is timed
        t0 = time.perf_counter()
        r = tr.evaluate(a.scalar_episodes)
        dt = time.perf_counter() - t0
        out["rows"].append(dict(algo=algo, path="episode-loop (B=1, host round trip per step)", episodes=a.scalar_episodes,
                                s_per_eval=round(dt, 5), env_steps_per_s=round(a.scalar_episodes * a.len / dt, 1)))
        if algo == "cpq":  # CPU oracle policy + numpy env (what the reference's evaluate() does, minus torch overhead)
            from oracle.osrl_oracle import OracleCPQ, rollout
            sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
            o = OracleCPQ(sd, max_action=1.0, sample_action_num=10, gamma=0.99, tau=0.005, beta=0.5, qc_scalar=1.5,
                          cost_limit=10, episode_len=a.len, actor_lr=1e-4, critic_lr=1e-3, alpha_lr=1e-4, vae_lr=1e-3)
            t0 = time.perf_counter()
            for _ in range(a.scalar_episodes):
                rollout(lambda ob: o.act(ob[None])[0], env, a.len)
            dt = time.perf_counter() - t0
            out["rows"].append(dict(algo=algo, path="cpu oracle policy + numpy env", episodes=a.scalar_episodes,
                                    s_per_eval=round(dt, 5), env_steps_per_s=round(a.scalar_episodes * a.len / dt, 1)))
    # CDT at the C5 architecture (obs 11, act 3, T=20, E=256, 3 layers, 8 heads), 100-step episodes
    from osrl_amd.algorithms import CDT, CDTTrainer
    torch.manual_seed(0)
    EL = 100
    m = CDT(11, 3, 1.0, seq_len=20, episode_len=EL, embedding_dim=256, num_layers=3, num_heads=8, use_rew=True,
            use_cost=True, cost_transform=True, stochastic=True, device=dev)
    tr = CDTTrainer(m, None, DummyLogger(), device=dev)
    env = SyntheticSafeEnv(11, 3, EL, seed=1, init_noise=0.5)
    for E in (1, 64, 256):
        tr.env = VecSyntheticSafeEnv(env, E, dev)
        tr.evaluate(E, 30.0, 5.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = tr.evaluate(E, 30.0, 5.0)
        dt = time.perf_counter() - t0
        out["rows"].append(dict(algo="cdt", path="batched-graph", episodes=E, episode_len=EL, s_per_eval=round(dt, 5),
                                env_steps_per_s=round(E * EL / dt, 1), us_per_env_step_launch=round(dt / EL * 1e6, 2),
                                mean_return=round(r[0], 4)))
    tr.env = env
    t0 = time.perf_counter()
    tr.evaluate(1, 30.0, 5.0)
    dt = time.perf_counter() - t0
    out["rows"].append(dict(algo="cdt", path="episode-loop (B=1, host round trip per step)", episodes=1, episode_len=EL,
                            s_per_eval=round(dt, 5), env_steps_per_s=round(EL / dt, 1)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
