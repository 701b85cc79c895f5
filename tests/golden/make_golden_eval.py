#!/usr/bin/env python3
"""Generate tests/golden/eval_rollouts.npz: the metric's "cost-return gap vs ref" reference side.

The REFERENCE trainers' own ``rollout()`` loops (imported from /root/reference, never copied) drive the build-owned
synthetic environment (osrl_amd/common/synthetic_env.py, numpy) with the reference models' ``act()`` as the policy,
from E seeded initial states per case; the per-episode (return, cost, length) are the fixture.  The GPU tests run
the same weights through the batched on-device ``evaluate()`` and compare episode by episode
(tests/test_gpu_data_eval.py::test_evaluate_cost_return_gap_vs_reference).  Build container only:

    python tests/golden/make_golden_eval.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from cases import BEARL_CASES, CASES, CDT_CASES, COPTIDICE_CASES, make_cdt_params  # noqa: E402
from make_golden import REF, NoiseQueue, _install_stubs, build  # noqa: E402

EVAL = dict(episodes=12, episode_len=25, env_seed=1, init_noise=0.7, base_seed=100, cost_scale=2.0)
EVAL_CASES = ["bc_small", "cpq_small", "bcql_small", "bearl_lap", "coptidice_small"]


class SeededEnv:
    """The reference's rollout() calls env.reset() without arguments; this hands it episode ``k``'s initial state."""

    def __init__(self, env, seed):
        self.env, self.seed = env, seed

    def reset(self):
        return self.env.reset(seed=self.seed)

    def step(self, a):
        return self.env.step(a)


def main():
    Logger = _install_stubs()
    sys.path.insert(0, REF)
    import torch
    import osrl.algorithms as algos
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv  # numpy only; no HIP library is touched
    nq = NoiseQueue(torch)
    nq.install()
    allc = {**CASES, **BEARL_CASES, **COPTIDICE_CASES}
    E, EL = EVAL["episodes"], EVAL["episode_len"]
    out = {"meta": np.array([f"torch={torch.__version__}", f"numpy={np.__version__}", repr(EVAL)])}
    for name in EVAL_CASES:
        c = allc[name]
        m, tr, lg = build(c, torch, algos, Logger)
        m.episode_len = EL
        tr.cost_scale = EVAL["cost_scale"] if c.algo != "bc" else 1.0
        env = SyntheticSafeEnv(c.od, c.ad, 50, seed=EVAL["env_seed"], init_noise=EVAL["init_noise"])
        z = np.random.RandomState(7).randn(E, 2 * c.ad).astype(np.float32)  # BCQ-L: one decode noise per episode
        res = []
        for e in range(E):
            tr.env = SeededEnv(env, EVAL["base_seed"] + e)
            if c.algo == "bcql":
                nq.q = [("z", z[e][None].copy()) for _ in range(EL)]
            r, n, cst = tr.rollout()
            nq.q = []
            res.append((r, cst, n))
        out[name] = np.array(res, np.float64)
        if c.algo == "bcql":
            out[name + "_z"] = z
        print(name, "mean ret/cost/len", out[name].mean(0))
    # CDT: CDTTrainer.rollout (cdt.py:436-518)
    c = CDT_CASES["cdt_small"]
    m = algos.CDT(c.od, c.ad, 1.0, seq_len=c.T, episode_len=EL, embedding_dim=c.E, num_layers=c.layers,
                  num_heads=c.heads, attention_dropout=0.0, residual_dropout=0.0, embedding_dropout=0.0, time_emb=True,
                  use_rew=True, use_cost=True, cost_transform=c.cost_transform, action_head_layers=1, cost_prefix=False,
                  stochastic=c.stochastic, init_temperature=0.1, target_entropy=-c.ad)
    sd = make_cdt_params(c)
    te = sd["timestep_emb.weight"]
    need = EL + c.T
    if te.shape[0] < need:  # the fixture's table was sized for episode_len 20: tile it (deterministic) to EL + T rows
        sd["timestep_emb.weight"] = np.concatenate([te] * (need // te.shape[0] + 1))[:need]
    else:
        sd["timestep_emb.weight"] = te[:need]
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    tr = algos.CDTTrainer(m, None, Logger(), reward_scale=0.1, cost_scale=EVAL["cost_scale"])
    m.eval()
    env = SyntheticSafeEnv(c.od, c.ad, 50, seed=EVAL["env_seed"], init_noise=EVAL["init_noise"])
    res = []
    for e in range(E):
        r, n, cst = tr.rollout(m, SeededEnv(env, EVAL["base_seed"] + e), 30.0, 5.0)
        res.append((r, cst, n))
    out["cdt_small"] = np.array(res, np.float64)
    print("cdt_small mean ret/cost/len", out["cdt_small"].mean(0))
    np.savez_compressed(os.path.join(HERE, "eval_rollouts.npz"), **out)


if __name__ == "__main__":
    main()
