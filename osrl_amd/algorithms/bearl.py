"""BEAR-Lagrangian on MI355X behind the reference's API (osrl/algorithms/bearl.py)."""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from ..common.logger import store_stats
from ..common.net import VAE, EnsembleDoubleQCritic, SquashedGaussianMLPActor, bind_group, plan_group
from ..engine.core import FlatGroup, require_cuda
from .bcql import LagrangianPIDController


class BEARL(nn.Module):
    """bearl.py:15-126: squashed-Gaussian actor, twin Q / Qc ensembles, VAE behaviour model, MMD support constraint
    with a dual variable ``log_alpha`` and a PID Lagrangian on the cost critic."""

    def __init__(self, state_dim: int, action_dim: int, max_action: float, a_hidden_sizes: list = [128, 128],
                 c_hidden_sizes: list = [128, 128], vae_hidden_sizes: int = 64, sample_action_num: int = 10,
                 gamma: float = 0.99, tau: float = 0.005, beta: float = 0.5, lmbda: float = 0.75,
                 mmd_sigma: float = 50, target_mmd_thresh: float = 0.05, num_samples_mmd_match: int = 10,
                 PID: list = [0.1, 0.003, 0.001], kernel: str = "gaussian", num_q: int = 1, num_qc: int = 1,
                 cost_limit: int = 10, episode_len: int = 300, start_update_policy_step: int = 20_000,
                 device: str = "cuda"):
        super().__init__()
        if kernel not in ("gaussian", "laplacian"):
            raise ValueError(f"kernel {kernel!r}: the reference knows 'gaussian' and 'laplacian' (bearl.py:234-241)")
        self.state_dim, self.action_dim, self.max_action = state_dim, action_dim, max_action
        self.latent_dim = self.action_dim * 2
        self.a_hidden_sizes, self.c_hidden_sizes = list(a_hidden_sizes), list(c_hidden_sizes)
        self.vae_hidden_sizes = vae_hidden_sizes
        self.sample_action_num = sample_action_num
        self.gamma, self.tau, self.beta, self.lmbda = gamma, tau, beta, lmbda
        self.mmd_sigma, self.target_mmd_thresh = mmd_sigma, target_mmd_thresh
        self.num_samples_mmd_match = num_samples_mmd_match
        self.start_update_policy_step = start_update_policy_step
        self.KP, self.KI, self.KD = PID
        self.kernel = kernel
        self.num_q, self.num_qc = num_q, num_qc
        self.cost_limit, self.episode_len = cost_limit, episode_len
        self.device = str(device)
        dev = require_cuda(device)

        # creation order of bearl.py:97-109 (actor, critic, cost_critic, vae) => same init under the same seed
        self.actor = SquashedGaussianMLPActor(state_dim, action_dim, self.a_hidden_sizes, nn.ReLU)
        self.critic = EnsembleDoubleQCritic(state_dim, action_dim, self.c_hidden_sizes, nn.ReLU, num_q=num_q)
        self.cost_critic = EnsembleDoubleQCritic(state_dim, action_dim, self.c_hidden_sizes, nn.ReLU, num_q=num_qc)
        self.vae = VAE(state_dim, action_dim, vae_hidden_sizes, self.latent_dim, max_action, self.device)
        self.actor_old = deepcopy(self.actor)
        self.critic_old = deepcopy(self.critic)
        self.cost_critic_old = deepcopy(self.cost_critic)
        for m in (self.actor_old, self.critic_old, self.cost_critic_old):
            m.eval()

        self.groups: Dict[str, FlatGroup] = {}
        for name, with_tgt in (("actor", True), ("critic", True), ("cost_critic", True), ("vae", False)):
            g = FlatGroup(name, dev, with_target=with_tgt)
            plan_group(g, name, getattr(self, name))
            g.finalize()
            bind_group(g, name, getattr(self, name), getattr(self, name + "_old") if with_tgt else None)
            self.groups[name] = g

        self.log_alpha = torch.zeros(1, dtype=torch.float32, device=dev)  # bearl.py:110: a plain tensor
        self.qc_thres = cost_limit * (1 - self.gamma ** self.episode_len) / (1 - self.gamma) / self.episode_len
        self.pid_state = torch.zeros(2, dtype=torch.float32, device=dev)
        self.controller = LagrangianPIDController(self.KP, self.KI, self.KD, self.qc_thres, self.pid_state)
        self._engine = None
        self._lrs: Optional[dict] = None
        self.alpha_lr = 0.0

    @property
    def n_train_steps(self) -> int:
        """bearl.py:95,268: completed actor updates (= the device step counter)."""
        from ..common.checkpoint import train_step_count
        return train_step_count(self)

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

    def setup_optimizers(self, actor_lr, critic_lr, vae_lr, alpha_lr):
        """bearl.py:314-321 (note the argument order: vae_lr before alpha_lr)."""
        self._lrs = dict(actor=actor_lr, critic=critic_lr, cost_critic=critic_lr, vae=vae_lr)
        self.alpha_lr = alpha_lr

    def engine(self, batch_size: int, **kw):
        from ..common.checkpoint import engine_handoff
        from ..engine.bearl import BEARLEngine
        if self._engine is None or self._engine.B != batch_size or kw:
            if self._lrs is None:
                raise RuntimeError("call setup_optimizers() (or build a BEARLTrainer) before training")
            old, self._engine = self._engine, BEARLEngine(self, batch_size, **kw)
            engine_handoff(self, self._engine, old)
        return self._engine

    def sync_weight(self):
        """bearl.py:329-335.  Fused into each group's optimizer kernel inside train_one_step."""
        return None

    @torch.no_grad()
    def act(self, obs: np.ndarray, deterministic: bool = False, with_logprob: bool = False):
        """bearl.py:337-350: single observation -> (max_action * tanh(u), logp)."""
        if getattr(self, "_fast", None) is None:
            from ..common.net import actor_head_desc
            from ..engine.act import FastPolicy
            self._fast = FastPolicy("gauss", self.device, self.state_dim, self.action_dim, actor_head_desc(self.actor),
                                    max_action=self.max_action)
        return self._fast.act(obs, deterministic)


class BEARLTrainer:
    """bearl.py:353-455."""

    def __init__(self, model: BEARL, env=None, logger=None, actor_lr: float = 1e-3, critic_lr: float = 1e-3,
                 alpha_lr: float = 1e-3, vae_lr: float = 1e-3, reward_scale: float = 1.0, cost_scale: float = 1.0,
                 device="cuda", stats_mode: str = "lazy", use_graph: bool = True):
        self.model, self.logger, self.env = model, logger, env
        self.reward_scale, self.cost_scale, self.device = reward_scale, cost_scale, device
        self.stats_mode, self.use_graph = stats_mode, use_graph
        self.model.setup_optimizers(actor_lr, critic_lr, vae_lr, alpha_lr)

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done, noise=None):
        """bearl.py:393-417 (vae -> critic -> cost critic -> actor -> Polyak).  ``noise``: optional dict of explicit
        standard-normal tensors (oracle/bearl_oracle.py header) for seeded parity; else drawn on device."""
        eng = self.model.engine(observations.shape[0])
        eng.step(observations, next_observations, actions, rewards, costs, done, noise=noise,
                 use_graph=self.use_graph and noise is None)
        store_stats(self.logger, eng.st, self.stats_mode)

    def evaluate(self, eval_episodes):
        """bearl.py:419-432.  A ``VecSyntheticSafeEnv`` as ``self.env`` runs the episodes as one batch on device (the
        deterministic policy max_action * tanh(mu) is CPQ's, engine/rollout.py)."""
        from ..common.synthetic_env import VecSyntheticSafeEnv
        if isinstance(self.env, VecSyntheticSafeEnv):
            from ..engine.rollout import evaluate_batched
            r, c, n = evaluate_batched(self, "cpq", eval_episodes, self.cost_scale)
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
        """bearl.py:434-455."""
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
