"""Helpers to drive the oracle over a tests/cases.py Case (tests only)."""
import os

import numpy as np

from cases import Case, dice_stds, hyper, make_batch, make_noise, make_params
from oracle.bearl_oracle import OracleBEARL
from oracle.coptidice_oracle import OracleCOptiDICE
from oracle.osrl_oracle import OracleBC, OracleBCQL, OracleCPQ

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def build_oracle(c: Case, dtype=np.float32):
    hp = hyper(c)
    sd = make_params(c)
    if c.algo == "bc":
        return OracleBC(sd, c.max_action, hp["actor_lr"], dtype=dtype)
    if c.algo == "cpq":
        return OracleCPQ(sd, max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"],
                         tau=hp["tau"], beta=hp["beta"], qc_scalar=hp["qc_scalar"],
                         cost_limit=c.cost_limit, episode_len=c.episode_len, actor_lr=hp["actor_lr"],
                         critic_lr=hp["critic_lr"], alpha_lr=hp["alpha_lr"], vae_lr=hp["vae_lr"],
                         dtype=dtype)
    if c.algo == "coptidice":
        ostd, astd = dice_stds(c)
        return OracleCOptiDICE(sd, max_action=c.max_action, f_type=hp["f_type"],
                               init_state_propotion=hp["init_state_propotion"], observations_std=ostd,
                               actions_std=astd, gamma=hp["gamma"], alpha=hp["alpha"],
                               cost_ub_epsilon=hp["cost_ub_epsilon"], cost_limit=c.cost_limit,
                               episode_len=c.episode_len, actor_lr=hp["actor_lr"], critic_lr=hp["critic_lr"],
                               scalar_lr=hp["scalar_lr"], dtype=dtype)
    if c.algo == "bearl":
        return OracleBEARL(sd, max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"], tau=hp["tau"],
                           beta=hp["beta"], lmbda=hp["lmbda"], mmd_sigma=hp["mmd_sigma"],
                           target_mmd_thresh=hp["target_mmd_thresh"], num_samples_mmd_match=hp["M"],
                           PID_gains=hp["PID"], kernel=hp["kernel"], cost_limit=c.cost_limit,
                           episode_len=c.episode_len, start_update_policy_step=hp["start"], actor_lr=hp["actor_lr"],
                           critic_lr=hp["critic_lr"], alpha_lr=hp["alpha_lr"], vae_lr=hp["vae_lr"], dtype=dtype)
    return OracleBCQL(sd, max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"],
                      tau=hp["tau"], phi=hp["phi"], lmbda=hp["lmbda"], beta=hp["beta"],
                      PID_gains=hp["PID"], cost_limit=c.cost_limit, episode_len=c.episode_len,
                      actor_lr=hp["actor_lr"], critic_lr=hp["critic_lr"], vae_lr=hp["vae_lr"],
                      dtype=dtype)


def oracle_step(o, c: Case, step: int):
    b = make_batch(c)
    if c.algo == "bc":
        return o.train_one_step(b["observations"], b["actions"])
    if c.algo == "coptidice":
        return o.train_one_step(b["observations"], b["next_observations"], b["actions"], b["rewards"], b["costs"],
                                b["done"], b["is_init"], make_noise(c, step))
    return o.train_one_step(b["observations"], b["next_observations"], b["actions"], b["rewards"],
                            b["costs"], b["done"], make_noise(c, step))
