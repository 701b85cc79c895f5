"""Build osrl_amd models/trainers from a tests/cases.py Case (GPU tests only)."""
import numpy as np
import torch

from cases import Case, dice_stds, hyper, make_batch, make_noise, make_params


def build_gpu(c: Case, device="cuda:0", **trainer_kw):
    from osrl_amd.algorithms import (BC, BCQL, BEARL, CPQ, BCQLTrainer, BCTrainer, BEARLTrainer, COptiDICE,
                                     COptiDICETrainer, CPQTrainer)
    from osrl_amd.common.logger import DummyLogger
    hp = hyper(c)
    lg = DummyLogger()
    kw = dict(stats_mode="sync", use_graph=False)
    kw.update(trainer_kw)
    if c.algo == "bc":
        m = BC(c.od, c.ad, c.max_action, c.hidden, c.episode_len, device=device)
        tr = BCTrainer(m, None, lg, actor_lr=hp["actor_lr"], device=device, **kw)
    elif c.algo == "cpq":
        m = CPQ(c.od, c.ad, c.max_action, c.hidden, c.hidden, c.vae_hidden, c.N, hp["gamma"], hp["tau"],
                hp["beta"], c.num_q, c.num_qc, hp["qc_scalar"], c.cost_limit, c.episode_len, device=device)
        tr = CPQTrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["alpha_lr"], hp["vae_lr"],
                        device=device, **kw)
    elif c.algo == "coptidice":
        ostd, astd = dice_stds(c)
        m = COptiDICE(c.od, c.ad, c.max_action, hp["f_type"], hp["init_state_propotion"], ostd, astd, c.hidden, c.hidden,
                      hp["gamma"], hp["alpha"], hp["cost_ub_epsilon"], c.num_q, c.num_qc, c.cost_limit, c.episode_len,
                      device=device)
        tr = COptiDICETrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["scalar_lr"], device=device, **kw)
    elif c.algo == "bearl":
        m = BEARL(c.od, c.ad, c.max_action, c.hidden, c.hidden, c.vae_hidden, c.N, hp["gamma"], hp["tau"], hp["beta"],
                  hp["lmbda"], hp["mmd_sigma"], hp["target_mmd_thresh"], hp["M"], list(hp["PID"]), hp["kernel"],
                  c.num_q, c.num_qc, c.cost_limit, c.episode_len, hp["start"], device=device)
        tr = BEARLTrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["alpha_lr"], hp["vae_lr"], device=device,
                          **kw)
    else:
        m = BCQL(c.od, c.ad, c.max_action, c.hidden, c.hidden, c.vae_hidden, c.N, hp["gamma"], hp["tau"],
                 hp["phi"], hp["lmbda"], hp["beta"], list(hp["PID"]), c.num_q, c.num_qc, c.cost_limit,
                 c.episode_len, device=device)
        tr = BCQLTrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["vae_lr"], device=device, **kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in make_params(c).items()}
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m, tr, lg


def gpu_batch(c: Case, device="cuda:0"):
    return {k: torch.from_numpy(v).to(device) for k, v in make_batch(c).items()}


def gpu_step(tr, c: Case, b, step: int, with_noise=True):
    if c.algo == "bc":
        tr.train_one_step(b["observations"], b["actions"])
    elif c.algo == "coptidice":
        nz = None
        if with_noise:
            nz = {k: torch.from_numpy(v).to(b["observations"].device) for k, v in make_noise(c, step).items()}
        tr.train_one_step([b[k] for k in ("observations", "next_observations", "actions", "rewards", "costs", "done",
                                          "is_init")], noise=nz)
    else:
        nz = None
        if with_noise:
            nz = {k: torch.from_numpy(v).to(b["observations"].device) for k, v in make_noise(c, step).items()}
        tr.train_one_step(b["observations"], b["next_observations"], b["actions"], b["rewards"], b["costs"],
                          b["done"], noise=nz)
