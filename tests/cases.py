"""Deterministic (numpy RandomState) inputs shared by the golden generator
(tests/golden/make_golden.py), the oracle tests and the GPU parity tests.

Weights follow nn.Linear's default distribution U(+-1/sqrt(fan_in)) but are drawn
from numpy so that every machine regenerates bit-identical tensors; the golden
generator loads them into the reference via ``load_state_dict``.
State-dict key layouts are the reference's.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np


@dataclass
class Case:
    name: str
    algo: str  # "bc" | "cpq" | "bcql"
    od: int
    ad: int
    B: int
    hidden: List[int]
    vae_hidden: int = 0
    N: int = 10
    num_q: int = 2
    num_qc: int = 2
    steps: int = 10
    max_action: float = 1.0
    episode_len: int = 300
    cost_limit: float = 10.0
    seed: int = 0
    hp: Dict[str, float] = field(default_factory=dict)


CASES: Dict[str, Case] = {c.name: c for c in [
    Case("bc_small", "bc", od=8, ad=2, B=32, hidden=[32, 32], steps=10),
    Case("bc_c1", "bc", od=8, ad=2, B=256, hidden=[256, 256], steps=3),
    Case("cpq_small", "cpq", od=5, ad=2, B=16, hidden=[32, 32], vae_hidden=48, N=4, steps=10,
         episode_len=1000),
    Case("cpq_odd", "cpq", od=7, ad=3, B=24, hidden=[48, 32], vae_hidden=40, N=3, num_q=1, num_qc=3,
         steps=3, max_action=2.0),
    Case("cpq_wide", "cpq", od=76, ad=2, B=64, hidden=[256, 256], vae_hidden=400, N=10, steps=1,
         episode_len=1000),
    Case("bcql_small", "bcql", od=6, ad=3, B=16, hidden=[32, 32], vae_hidden=48, N=4, steps=10,
         episode_len=200),
    Case("bcql_pid", "bcql", od=4, ad=2, B=16, hidden=[24, 24], vae_hidden=32, N=3, num_q=1, num_qc=2,
         steps=5, episode_len=200, cost_limit=-4.0, max_action=1.5),
    Case("bcql_wide", "bcql", od=33, ad=8, B=32, hidden=[256, 256], vae_hidden=400, N=10, steps=1,
         episode_len=200),
]}


def _linear(rs, out_f, in_f):
    k = 1.0 / np.sqrt(in_f)
    return (rs.uniform(-k, k, (out_f, in_f)).astype(np.float32),
            rs.uniform(-k, k, (out_f,)).astype(np.float32))


def _seq(rs, sd, prefix, sizes):
    for i in range(len(sizes) - 1):
        w, b = _linear(rs, sizes[i + 1], sizes[i])
        sd[f"{prefix}.{2 * i}.weight"], sd[f"{prefix}.{2 * i}.bias"] = w, b


def _named(rs, sd, name, out_f, in_f):
    sd[name + ".weight"], sd[name + ".bias"] = _linear(rs, out_f, in_f)


def _vae(rs, sd, od, ad, V):
    L = 2 * ad
    _named(rs, sd, "vae.e1", V, od + ad)
    _named(rs, sd, "vae.e2", V, V)
    _named(rs, sd, "vae.mean", L, V)
    _named(rs, sd, "vae.log_std", L, V)
    _named(rs, sd, "vae.d1", V, od + L)
    _named(rs, sd, "vae.d2", V, V)
    _named(rs, sd, "vae.d3", ad, V)


def make_params(c: Case) -> "OrderedDict[str, np.ndarray]":
    """Initial state_dict (targets = copies, as deepcopy does in cpq.py:95-100)."""
    rs = np.random.RandomState(1000 + c.seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    if c.algo == "bc":
        _seq(rs, sd, "actor.pi", [c.od] + c.hidden + [c.ad])
        return sd
    if c.algo == "cpq":
        _seq(rs, sd, "actor.net", [c.od] + c.hidden)
        _named(rs, sd, "actor.mu_layer", c.ad, c.hidden[-1])
        _named(rs, sd, "actor.log_std_layer", c.ad, c.hidden[-1])
        for i in range(c.num_q):
            _seq(rs, sd, f"critic.q_nets.{i}", [c.od + c.ad] + c.hidden + [1])
        _vae(rs, sd, c.od, c.ad, c.vae_hidden)
        for i in range(c.num_qc):
            _seq(rs, sd, f"cost_critic.q_nets.{i}", [c.od + c.ad] + c.hidden + [1])
    elif c.algo == "bcql":
        _seq(rs, sd, "actor.pi", [c.od + c.ad] + c.hidden + [c.ad])
        for grp, n in (("critic", c.num_q), ("cost_critic", c.num_qc)):
            for which in ("q1_nets", "q2_nets"):
                for i in range(n):
                    _seq(rs, sd, f"{grp}.{which}.{i}", [c.od + c.ad] + c.hidden + [1])
        _vae(rs, sd, c.od, c.ad, c.vae_hidden)
    else:
        raise ValueError(c.algo)
    for k in list(sd.keys()):
        for src in ("actor", "critic", "cost_critic"):
            if k.startswith(src + "."):
                sd[src + "_old" + k[len(src):]] = sd[k].copy()
    return sd


def make_batch(c: Case) -> Dict[str, np.ndarray]:
    """Synthetic transitions as SURVEY.md 8d: obs~N(0,1), act~U(-1,1)*max_a, rew~N(0,1),
    cost~Bern(.1), done~Bern(.01) (a few forced so the done branch is exercised)."""
    rs = np.random.RandomState(2000 + c.seed)
    f = np.float32
    b = dict(
        observations=rs.randn(c.B, c.od).astype(f),
        next_observations=rs.randn(c.B, c.od).astype(f),
        actions=(rs.uniform(-1, 1, (c.B, c.ad)) * c.max_action).astype(f),
        rewards=rs.randn(c.B).astype(f),
        costs=(rs.uniform(size=c.B) < 0.1).astype(f),
        done=(rs.uniform(size=c.B) < 0.01).astype(f),
    )
    b["done"][:: max(c.B // 3, 1)] = 1.0
    return b


def noise_shapes(c: Case):
    """Per-step noise tensors in the reference's RNG draw order."""
    B, ad, N = c.B, c.ad, c.N
    if c.algo == "cpq":
        return [("eps_vae", (B, 2 * ad)), ("eps_next_c", (B, ad)), ("eps_next_cc", (B, ad)),
                ("eps_pi_unused", (B, ad)), ("eps_ood", (N, B, ad)), ("eps_vae_ood", (N * B, 2 * ad)),
                ("eps_actor", (B, ad))]
    if c.algo == "bcql":
        return [("eps_vae", (B, 2 * ad)), ("z_c", (N * B, 2 * ad)), ("z_cc", (N * B, 2 * ad)),
                ("z_actor", (B, 2 * ad))]
    return []


def make_noise(c: Case, step: int) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(3000 + 97 * c.seed + step)
    return {k: rs.randn(*s).astype(np.float32) for k, s in noise_shapes(c)}


def hyper(c: Case) -> Dict[str, float]:
    """Train-config defaults (cpq_configs.py:31-50, bcql_configs.py:31-51, bc_configs.py)."""
    if c.algo == "bc":
        return dict(actor_lr=1e-3)
    if c.algo == "cpq":
        return dict(actor_lr=1e-4, critic_lr=1e-3, alpha_lr=1e-4, vae_lr=1e-3, gamma=0.99, tau=0.005,
                    beta=0.5, qc_scalar=1.5)
    return dict(actor_lr=1e-3, critic_lr=1e-3, vae_lr=1e-3, gamma=0.99, tau=0.005, beta=0.5,
                phi=0.05, lmbda=0.75, PID=(0.1, 0.003, 0.001))
