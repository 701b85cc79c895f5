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
    elif c.algo == "coptidice":  # creation order of coptidice.py:98-111: actor, nu_network, chi_network (no targets)
        _seq(rs, sd, "actor.net", [c.od] + c.hidden)
        _named(rs, sd, "actor.mu_layer", c.ad, c.hidden[-1])
        _named(rs, sd, "actor.log_std_layer", c.ad, c.hidden[-1])
        for grp, n in (("nu_network", c.num_q), ("chi_network", c.num_qc)):
            for i in range(n):
                _seq(rs, sd, f"{grp}.q_nets.{i}", [c.od] + c.hidden + [1])
        return sd
    elif c.algo == "bearl":  # creation order of bearl.py:97-109: actor, critic, cost_critic, vae
        _seq(rs, sd, "actor.net", [c.od] + c.hidden)
        _named(rs, sd, "actor.mu_layer", c.ad, c.hidden[-1])
        _named(rs, sd, "actor.log_std_layer", c.ad, c.hidden[-1])
        for grp, n in (("critic", c.num_q), ("cost_critic", c.num_qc)):
            for which in ("q1_nets", "q2_nets"):
                for i in range(n):
                    _seq(rs, sd, f"{grp}.{which}.{i}", [c.od + c.ad] + c.hidden + [1])
        _vae(rs, sd, c.od, c.ad, c.vae_hidden)
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
    if c.algo == "coptidice":  # TransitionDataset(state_init=True) adds the initial-state flag (dataset.py:817-820)
        b["is_init"] = (rs.uniform(size=c.B) < 0.15).astype(f)
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
    if c.algo == "coptidice":  # oracle/coptidice_oracle.py header
        # the third draw is the actor's rsample inside forward(deterministic=False) (coptidice.py:207): only the
        # distribution is used, the sample is discarded -- result-irrelevant but it consumes RNG
        return [("obs_eps", (B, c.od)), ("act_eps", (B, ad)), ("eps_pi_unused", (B, ad))]
    if c.algo == "bearl":  # oracle/bearl_oracle.py header
        M = int(c.hp.get("M", 10))
        return [("eps_vae", (B, 2 * ad)), ("eps_c", (N * B, ad)), ("eps_cc", (N * B, ad)),
                ("z_mmd", (B, M, 2 * ad)), ("eps_pi", (B * M, ad))]
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
    if c.algo == "coptidice":  # coptidice_configs.py:31-47
        d = dict(actor_lr=1e-4, critic_lr=1e-4, scalar_lr=1e-4, gamma=0.99, alpha=0.5, cost_ub_epsilon=0.01,
                 f_type="softchi", init_state_propotion=0.15)
        d.update(c.hp)
        return d
    if c.algo == "bearl":  # bearl_configs.py:31-57
        d = dict(actor_lr=1e-3, critic_lr=1e-3, vae_lr=1e-3, alpha_lr=1e-3, gamma=0.99, tau=0.005, beta=0.5, lmbda=0.75,
                 mmd_sigma=50.0, target_mmd_thresh=0.05, M=10, start=0, kernel="gaussian", PID=(0.1, 0.003, 0.001))
        d.update(c.hp)
        return d
    return dict(actor_lr=1e-3, critic_lr=1e-3, vae_lr=1e-3, gamma=0.99, tau=0.005, beta=0.5,
                phi=0.05, lmbda=0.75, PID=(0.1, 0.003, 0.001))


# BEAR-Lagrangian; kept apart from CASES so the older fixtures need no regeneration
BEARL_CASES: Dict[str, Case] = {c.name: c for c in [
    Case("bearl_small", "bearl", od=6, ad=3, B=16, hidden=[32, 32], vae_hidden=48, N=4, steps=10, episode_len=200,
         hp=dict(M=5, mmd_sigma=2.0)),
    Case("bearl_lap", "bearl", od=4, ad=2, B=16, hidden=[24, 24], vae_hidden=32, N=3, num_q=1, num_qc=2, steps=5,
         episode_len=200, cost_limit=-4.0, max_action=1.5, seed=1,
         hp=dict(M=4, kernel="laplacian", mmd_sigma=1.5, start=3, alpha_lr=0.05)),
    Case("bearl_wide", "bearl", od=33, ad=8, B=32, hidden=[256, 256], vae_hidden=400, N=10, steps=1, episode_len=200),
]}


# --------------------------------------------------------------------------- #
# CDT (osrl/algorithms/cdt.py) cases
# --------------------------------------------------------------------------- #
@dataclass
class CDTCase:
    name: str
    od: int
    ad: int
    B: int
    T: int
    E: int
    heads: int
    layers: int
    episode_len: int = 50
    steps: int = 5
    stochastic: bool = True
    cost_transform: bool = True
    warmup: int = 3
    clip: float = 0.25
    lr: float = 1e-4
    wd: float = 1e-4
    cost_w: float = 0.02
    state_w: float = 0.0
    seed: int = 0
    algo: str = "cdt"
    dropout: float = 0.0  # attention / residual / embedding dropout (cdt_configs.py:28-30 default 0.1)
    # constructor variants of cdt.py:45-70 (all non-default in the reference's configs)
    time_emb: bool = True
    use_rew: bool = True
    use_cost: bool = True
    add_cost_feat: bool = False
    mul_cost_feat: bool = False
    cat_cost_feat: bool = False
    head_layers: int = 1
    cost_prefix: bool = False

    @property
    def R(self) -> int:  # tokens per timestep (cdt.py:96-105)
        return 2 + int(self.use_rew) + int(self.use_cost)

    @property
    def S(self) -> int:  # sequence length the transformer sees (cdt.py:107-112)
        return self.R * self.T + int(self.cost_prefix)


CDT_CASES: Dict[str, CDTCase] = {c.name: c for c in [
    CDTCase("cdt_small", od=5, ad=3, B=6, T=4, E=16, heads=2, layers=2, episode_len=20, steps=5, state_w=0.1),
    CDTCase("cdt_det", od=4, ad=2, B=5, T=3, E=16, heads=4, layers=1, episode_len=12, steps=3, stochastic=False,
            cost_transform=False, clip=1e9, warmup=1),
    CDTCase("cdt_drop", od=5, ad=3, B=6, T=5, E=16, heads=2, layers=2, episode_len=20, steps=4, state_w=0.1,
            dropout=0.1, seed=3),
    CDTCase("cdt_mid", od=11, ad=3, B=16, T=10, E=128, heads=8, layers=3, episode_len=1000, steps=1, warmup=500),
    # ---- constructor variants
    CDTCase("cdt_v_norew", od=5, ad=3, B=6, T=4, E=16, heads=2, layers=2, episode_len=20, steps=4, state_w=0.1,
            use_rew=False, add_cost_feat=True, head_layers=2, seed=11),
    CDTCase("cdt_v_min", od=4, ad=2, B=5, T=3, E=16, heads=4, layers=1, episode_len=12, steps=3, stochastic=False,
            cost_transform=False, clip=1e9, warmup=1, time_emb=False, use_rew=False, use_cost=False, head_layers=3,
            seed=12),
    CDTCase("cdt_v_prefix", od=5, ad=3, B=6, T=5, E=16, heads=2, layers=2, episode_len=20, steps=4, state_w=0.1,
            cost_prefix=True, mul_cost_feat=True, cat_cost_feat=True, dropout=0.1, seed=13),
    CDTCase("cdt_v_prefix_det", od=4, ad=2, B=5, T=4, E=32, heads=4, layers=1, episode_len=12, steps=3,
            stochastic=False, cost_prefix=True, use_rew=False, cat_cost_feat=True, add_cost_feat=True, head_layers=2,
            seed=14),
]}


def make_cdt_params(c: CDTCase) -> "OrderedDict[str, np.ndarray]":
    """state_dict of the reference CDT (keys: SURVEY.md 8b).  Linear/Embedding ~ N(0,.02) as cdt.py:156-164;
    biases and LayerNorm affine get small random values so their gradient paths are exercised."""
    rs = np.random.RandomState(5000 + c.seed)
    f = np.float32
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    E = c.E

    def lin(name, out_f, in_f):
        sd[name + ".weight"] = (rs.randn(out_f, in_f) * 0.02).astype(f)
        sd[name + ".bias"] = (rs.randn(out_f) * 0.02).astype(f)

    def ln(name):
        sd[name + ".weight"] = (1 + rs.randn(E) * 0.05).astype(f)
        sd[name + ".bias"] = (rs.randn(E) * 0.05).astype(f)

    ln("emb_norm")
    ln("out_norm")
    if c.time_emb:
        sd["timestep_emb.weight"] = (rs.randn(c.episode_len + c.T, E) * 0.02).astype(f)
    lin("state_emb", E, c.od)
    lin("action_emb", E, c.ad)
    if c.use_cost:
        lin("cost_emb", E, 1)
    if c.use_rew:
        lin("return_emb", E, 1)
    if c.cost_prefix:
        lin("prefix_emb", E, 1)
    S = c.S
    for l in range(c.layers):
        pre = f"blocks.{l}"
        sd[pre + ".causal_mask"] = ~np.tril(np.ones((S, S))).astype(bool)
        ln(pre + ".norm1")
        ln(pre + ".norm2")
        sd[pre + ".attention.in_proj_weight"] = (rs.randn(3 * E, E) * 0.05).astype(f)
        sd[pre + ".attention.in_proj_bias"] = (rs.randn(3 * E) * 0.02).astype(f)
        lin(pre + ".attention.out_proj", E, E)
        lin(pre + ".mlp.0", 4 * E, E)
        lin(pre + ".mlp.2", E, 4 * E)
    Eh = 2 * E if c.cat_cost_feat else E  # cdt.py:125
    if c.stochastic:  # cdt.py:127-133
        if c.head_layers >= 2:
            lin("action_head.0", Eh, Eh)
            lin("action_head.2.mu", c.ad, Eh)
            lin("action_head.2.log_std", c.ad, Eh)
        else:
            lin("action_head.mu", c.ad, Eh)
            lin("action_head.log_std", c.ad, Eh)
    else:  # cdt.py:134-137: mlp([Eh] * head_layers + [ad], GELU, Identity)
        for i in range(c.head_layers - 1):
            lin(f"action_head.{2 * i}", Eh, Eh)
        lin(f"action_head.{2 * (c.head_layers - 1)}", c.ad, Eh)
    lin("state_pred_head", c.od, E)
    lin("cost_pred_head", 2, E)
    return sd


def cdt_drop_sites(c: CDTCase):
    """(key, shape) of every nn.Dropout draw of one CDT forward, in the reference's call order."""
    S = c.S
    out = [("emb", (c.B, S, c.E))]
    for l in range(c.layers):
        out += [(f"attn{l}", (c.B, c.heads, S, S)), (f"res1_{l}", (c.B, S, c.E)), (f"res2_{l}", (c.B, S, c.E))]
    return out


def make_cdt_drop(c: CDTCase, step: int) -> Dict[str, np.ndarray]:
    """Keep-multipliers (0 or 1/(1-p)) for train step ``step``."""
    rs = np.random.RandomState(7000 + 131 * c.seed + step)
    p = c.dropout
    return {k: ((rs.uniform(size=shp) >= p) / (1.0 - p)).astype(np.float32) for k, shp in cdt_drop_sites(c)}


def make_cdt_batch(c: CDTCase) -> Dict[str, np.ndarray]:
    """states~N(0,1), actions~U(-1,1), returns~U(0,100)*0.1, costs_return~U(0,20),
    time_steps=start+arange(T), mask ones with some rows tail-padded (zero-filled), costs~Bern(.1)."""
    rs = np.random.RandomState(6000 + c.seed)
    f = np.float32
    B, T = c.B, c.T
    start = rs.randint(0, c.episode_len, size=B)
    mask = np.ones((B, T), f)
    states = rs.randn(B, T, c.od).astype(f)
    actions = rs.uniform(-1, 1, (B, T, c.ad)).astype(f)
    returns = (rs.uniform(0, 100, (B, T)) * 0.1).astype(f)
    ctg = rs.uniform(0, 20, (B, T)).astype(f)
    costs = (rs.uniform(size=(B, T)) < 0.3).astype(f)
    for b in range(0, B, 3):  # tail padding like SequenceDataset.__prepare_sample (dataset.py:764-773)
        n = 1 + (b % (T - 1))
        mask[b, T - n:] = 0
        states[b, T - n:] = 0
        actions[b, T - n:] = 0
        returns[b, T - n:] = 0
        ctg[b, T - n:] = 0
        costs[b, T - n:] = 0
    return dict(states=states, actions=actions, returns=returns, costs_return=ctg,
                time_steps=(start[:, None] + np.arange(T)[None]).astype(np.int64), mask=mask,
                episode_cost=rs.uniform(0, 20, B).astype(f), costs=costs)


# --------------------------------------------------------------------------- #
# dataset ingestion: a small DSRL-shaped dataset with ragged episodes
# --------------------------------------------------------------------------- #
def make_ingest_dataset(seed: int = 0, n: int = 1500, od: int = 4, ad: int = 2, max_len: int = 60,
                        tail: int = 17) -> Dict[str, np.ndarray]:
    """Episodes of length 1..max_len ending in a timeout or (1 in 4) a terminal, one episode where both flags are
    set, a trailing partial episode of ``tail`` transitions without a done flag; costs ~ Bernoulli(0.25)."""
    rs = np.random.RandomState(seed)
    f = np.float32
    term, tout = np.zeros(n, f), np.zeros(n, f)
    i, k = 0, 0
    while True:
        L_ = int(rs.randint(1, max_len + 1))
        if i + L_ > n - tail:
            break
        i += L_
        if k % 4 == 3:
            term[i - 1] = 1
        else:
            tout[i - 1] = 1
        if k == 5:
            term[i - 1] = tout[i - 1] = 1
        k += 1
    return dict(observations=rs.randn(n, od).astype(f), next_observations=rs.randn(n, od).astype(f),
                actions=rs.uniform(-1, 1, (n, ad)).astype(f), rewards=rs.uniform(0, 2, n).astype(f),
                costs=(rs.uniform(size=n) < 0.25).astype(f), terminals=term, timeouts=tout)


def dice_stds(c: Case):
    """observations_std / actions_std as TransitionDataset.get_dataset_states returns them (dataset.py:827-829)."""
    rs = np.random.RandomState(5000 + c.seed)
    return (rs.uniform(0.5, 1.5, (1, c.od)).astype(np.float32), rs.uniform(0.3, 0.8, (1, c.ad)).astype(np.float32))


# COptiDICE.  num_q = num_nu, num_qc = num_chi.  Learning rates above the config's 1e-4 so that ten
# steps move the scalars and networks by more than the test tolerance.
COPTIDICE_CASES: Dict[str, Case] = {c.name: c for c in [
    Case("coptidice_small", "coptidice", od=6, ad=3, B=32, hidden=[32, 32], steps=10, episode_len=200,
         hp=dict(actor_lr=1e-3, critic_lr=1e-3, scalar_lr=1e-2)),
    Case("coptidice_chi2", "coptidice", od=5, ad=2, B=24, hidden=[24, 24], num_q=1, num_qc=3, steps=5, episode_len=200,
         cost_limit=40.0, seed=1, hp=dict(f_type="chi2", alpha=0.8, cost_ub_epsilon=0.0, actor_lr=1e-3, critic_lr=1e-3,
                                          scalar_lr=1e-2)),
    Case("coptidice_kl", "coptidice", od=4, ad=2, B=16, hidden=[16, 16], num_q=3, num_qc=1, steps=5, episode_len=100,
         seed=2, hp=dict(f_type="kl", cost_ub_epsilon=0.05, actor_lr=1e-3, critic_lr=1e-3, scalar_lr=1e-2)),
    Case("coptidice_wide", "coptidice", od=33, ad=8, B=512, hidden=[256, 256], steps=1, episode_len=300),
]}
