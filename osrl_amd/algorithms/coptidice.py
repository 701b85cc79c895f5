"""COptiDICE on MI355X behind the reference's API (osrl/algorithms/coptidice.py)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from ..common.logger import store_stats
from ..common.net import EnsembleQCritic, SquashedGaussianMLPActor, bind_group, plan_group
from ..engine.core import FlatGroup, require_cuda

F_TYPES = ("chi2", "softchi", "kl")


class COptiDICE(nn.Module):
    """coptidice.py:41-120: squashed-Gaussian actor, nu and chi state-value ensembles (``EnsembleQCritic`` with
    act_dim 0), scalar leaves ``tau`` / ``lmbda`` (plain tensors with their own Adam, not in ``state_dict``)."""

    def __init__(self, state_dim: int, action_dim: int, max_action: float, f_type: str, init_state_propotion: float,
                 observations_std: np.ndarray, actions_std: np.ndarray, a_hidden_sizes: list = [128, 128],
                 c_hidden_sizes: list = [128, 128], gamma: float = 0.99, alpha: float = 0.5,
                 cost_ub_epsilon: float = 0.01, num_nu: int = 1, num_chi: int = 1, cost_limit: int = 10,
                 episode_len: int = 300, device: str = "cuda"):
        super().__init__()
        if f_type not in F_TYPES:
            raise NotImplementedError(f"Not implemented f_fn: {f_type}")  # coptidice.py:36
        self.state_dim, self.action_dim, self.max_action = state_dim, action_dim, max_action
        self.f_type = f_type
        self.a_hidden_sizes, self.c_hidden_sizes = list(a_hidden_sizes), list(c_hidden_sizes)
        self.gamma, self.alpha, self.cost_ub_epsilon = gamma, alpha, cost_ub_epsilon
        self.num_nu, self.num_chi = num_nu, num_chi
        self.cost_limit, self.episode_len = cost_limit, episode_len
        self.init_state_propotion = float(init_state_propotion)
        self.device = str(device)
        dev = require_cuda(device)
        self.qc_thres = cost_limit * (1 - self.gamma ** self.episode_len) / (1 - self.gamma) / self.episode_len

        # creation order of coptidice.py:98-111 (actor, nu_network, chi_network)
        self.actor = SquashedGaussianMLPActor(state_dim, action_dim, self.a_hidden_sizes, nn.ReLU)
        self.nu_network = EnsembleQCritic(state_dim, 0, self.c_hidden_sizes, nn.ReLU, num_q=num_nu)
        self.chi_network = EnsembleQCritic(state_dim, 0, self.c_hidden_sizes, nn.ReLU, num_q=num_chi)
        self.groups: Dict[str, FlatGroup] = {}
        for name in ("actor", "nu_network", "chi_network"):
            g = FlatGroup(name, dev, with_target=False)
            plan_group(g, name, getattr(self, name))
            g.finalize()
            bind_group(g, name, getattr(self, name), None)
            self.groups[name] = g

        # {tau, m, v, lmbda, m, v}: raw leaves, both start at 1 (coptidice.py:96-97), with their Adam moments
        self.scalar_leaves = torch.tensor([1.0, 0.0, 0.0, 1.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        t = lambda a, n: torch.as_tensor(np.asarray(a, np.float32).reshape(-1), device=dev).contiguous().reshape(1, n)  # noqa: E731
        self.observations_std, self.actions_std = t(observations_std, state_dim), t(actions_std, action_dim)
        self._engine = None
        self._lrs: Optional[dict] = None
        self.scalar_lr = 0.0

    @property
    def tau(self) -> torch.Tensor:
        return self.scalar_leaves[0:1]

    @property
    def lmbda(self) -> torch.Tensor:
        return self.scalar_leaves[3:4]

    def repack(self) -> None:
        for g in self.groups.values():
            if g.device.type == "cuda":
                g.repack()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        if assign:
            raise RuntimeError("assign=True would detach parameters from their flat HBM groups")
        res = super().load_state_dict(state_dict, strict=strict)
        self.repack()
        return res

    def _apply(self, fn, *a, **k):
        raise RuntimeError("osrl_amd models are bound to their HIP device at construction (pass device=)")

    def setup_optimizers(self, actor_lr, critic_lr, scalar_lr):
        """coptidice.py:236-242."""
        self._lrs = dict(actor=actor_lr, nu_network=critic_lr, chi_network=critic_lr)
        self.scalar_lr = scalar_lr

    def engine(self, batch_size: int, **kw):
        from ..common.checkpoint import engine_handoff
        from ..engine.coptidice import COptiDICEEngine
        if self._engine is None or self._engine.B != batch_size or kw:
            if self._lrs is None:
                raise RuntimeError("call setup_optimizers() (or build a COptiDICETrainer) before training")
            old, self._engine = self._engine, COptiDICEEngine(self, batch_size, **kw)
            engine_handoff(self, self._engine, old)
        return self._engine

    def update(self, batch, noise=None, use_graph: bool = True):
        """coptidice.py:135-232: ``batch`` = (observations, next_observations, actions, rewards, costs, done,
        is_init).  Returns the engine; the statistics live in its device ring."""
        eng = self.engine(batch[0].shape[0])
        eng.step(*batch, noise=noise, use_graph=use_graph and noise is None)
        return eng

    @torch.no_grad()
    def act(self, obs: np.ndarray, deterministic: bool = False, with_logprob: bool = False):
        """coptidice.py:244-256: ``actor.forward`` directly -- tanh(u) WITHOUT max_action scaling."""
        if getattr(self, "_fast", None) is None:
            from ..common.net import actor_head_desc
            from ..engine.act import FastPolicy
            self._fast = FastPolicy("gauss", self.device, self.state_dim, self.action_dim, actor_head_desc(self.actor),
                                    max_action=1.0)
        return self._fast.act(obs, deterministic)


class COptiDICETrainer:
    """coptidice.py:259-321."""

    def __init__(self, model: COptiDICE, env=None, logger=None, actor_lr: float = 1e-3,
                 critic_lr: float = 1e-3, scalar_lr: float = 1e-3, reward_scale: float = 1.0, cost_scale: float = 1.0,
                 device="cuda", stats_mode: str = "lazy", use_graph: bool = True):
        self.model, self.logger, self.env = model, logger, env
        self.reward_scale, self.cost_scale, self.device = reward_scale, cost_scale, device
        self.stats_mode, self.use_graph = stats_mode, use_graph
        self.model.setup_optimizers(actor_lr, critic_lr, scalar_lr)

    def train_one_step(self, batch, noise=None):
        """coptidice.py:289-291.  ``noise``: optional {"obs_eps", "act_eps"} standard-normal tensors for seeded
        parity (oracle/coptidice_oracle.py header); else drawn on device."""
        eng = self.model.update(batch, noise=noise, use_graph=self.use_graph)
        store_stats(self.logger, eng.st, self.stats_mode)

    def evaluate(self, eval_episodes):
        """coptidice.py:293-306.  A ``VecSyntheticSafeEnv`` as ``self.env`` runs the episodes as one batch on device."""
        from ..common.synthetic_env import VecSyntheticSafeEnv
        if isinstance(self.env, VecSyntheticSafeEnv):
            from ..engine.rollout import evaluate_batched
            r, c, n = evaluate_batched(self, "dice", eval_episodes, self.cost_scale)
            return r / self.reward_scale, c / self.cost_scale, n
        self.model.eval()
        rets, costs, lens = [], [], []
        for _ in range(eval_episodes):
            r, l, c = self.rollout()
            rets.append(r); lens.append(l); costs.append(c)
        self.model.train()
        return np.mean(rets) / self.reward_scale, np.mean(costs) / self.cost_scale, np.mean(lens)

    @torch.no_grad()
    def rollout(self):
        """coptidice.py:308-321."""
        obs, info = self.env.reset()
        ep_ret, ep_cost, ep_len = 0.0, 0.0, 0
        for _ in range(self.model.episode_len):
            act, _ = self.model.act(obs, True, True)
            obs_next, reward, terminated, truncated, info = self.env.step(act)
            cost = info["cost"] * self.cost_scale
            obs = obs_next
            ep_ret += reward
            ep_len += 1
            ep_cost += cost
            if terminated or truncated:
                break
        return ep_ret, ep_len, ep_cost
