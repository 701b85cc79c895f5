"""CPQ on MI355X behind the reference's API (osrl/algorithms/cpq.py of liuzuxin/OSRL).

``CPQ`` keeps the reference's constructor, attribute names and ``state_dict`` keys
(cpq.py:38-105); ``CPQTrainer`` keeps ``train_one_step(observations, next_observations,
actions, rewards, costs, done)`` / ``evaluate`` / ``rollout`` (cpq.py:272-347).  The
arithmetic of the step is the fused HIP plan in ``osrl_amd/engine/cpq.py``.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from ..common.net import (VAE, EnsembleQCritic, SquashedGaussianMLPActor, bind_group, plan_group)
from ..engine.core import FlatGroup, require_cuda


class CPQ(nn.Module):
    """Constraints Penalized Q-Learning model container (reference cpq.py:13-105)."""

    def __init__(self,
                 state_dim: int,
                 action_dim: int,
                 max_action: float,
                 a_hidden_sizes: list = [128, 128],
                 c_hidden_sizes: list = [128, 128],
                 vae_hidden_sizes: int = 64,
                 sample_action_num: int = 10,
                 gamma: float = 0.99,
                 tau: float = 0.005,
                 beta: float = 1.5,
                 num_q: int = 1,
                 num_qc: int = 1,
                 qc_scalar: float = 1.5,
                 cost_limit: int = 10,
                 episode_len: int = 300,
                 device: str = "cuda"):
        super().__init__()
        self.a_hidden_sizes = list(a_hidden_sizes)
        self.c_hidden_sizes = list(c_hidden_sizes)
        self.vae_hidden_sizes = vae_hidden_sizes
        self.gamma, self.tau, self.beta = gamma, tau, beta
        self.cost_limit = cost_limit
        self.num_q, self.num_qc = num_q, num_qc
        self.qc_scalar = qc_scalar
        self.sample_action_num = sample_action_num
        self.state_dim, self.action_dim = state_dim, action_dim
        self.latent_dim = self.action_dim * 2
        self.episode_len = episode_len
        self.max_action = max_action
        self.device = str(device)
        dev = require_cuda(device)

        # same creation order as the reference (cpq.py:78-92) => same init under the same seed
        self.actor = SquashedGaussianMLPActor(state_dim, action_dim, self.a_hidden_sizes, nn.ReLU)
        self.critic = EnsembleQCritic(state_dim, action_dim, self.c_hidden_sizes, nn.ReLU, num_q=num_q)
        self.vae = VAE(state_dim, action_dim, vae_hidden_sizes, self.latent_dim, max_action, self.device)
        self.cost_critic = EnsembleQCritic(state_dim, action_dim, self.c_hidden_sizes, nn.ReLU, num_q=num_qc)
        self.actor_old = deepcopy(self.actor)
        self.critic_old = deepcopy(self.critic)
        self.cost_critic_old = deepcopy(self.cost_critic)
        for m in (self.actor_old, self.critic_old, self.cost_critic_old):
            m.eval()

        # flat HBM optimizer groups; parameters become views
        self.groups: Dict[str, FlatGroup] = {}
        for name, with_tgt in (("actor", True), ("critic", True), ("cost_critic", True), ("vae", False)):
            g = FlatGroup(name, dev, with_target=with_tgt)
            plan_group(g, name, getattr(self, name))
            g.finalize()
            bind_group(g, name, getattr(self, name), getattr(self, name + "_old") if with_tgt else None)
            self.groups[name] = g

        # scalar state stays on the device (cpq.py:93 keeps log_alpha as a plain tensor too)
        self.log_alpha = torch.zeros(1, dtype=torch.float32, device=dev)
        self.q_thres = cost_limit * (1 - self.gamma ** self.episode_len) / (1 - self.gamma) / self.episode_len
        self.qc_thres = qc_scalar * self.q_thres
        self._engine = None
        self._fast = None
        self._lrs: Optional[dict] = None

    def repack(self) -> None:
        """Refresh the fragment-ordered weight copies the kernels read; call after modifying parameters
        in place from outside the trainer (load_state_dict does it automatically)."""
        for g in self.groups.values():
            if g.device.type == "cuda":
                g.repack()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        if assign:
            raise RuntimeError("assign=True would detach parameters from their flat HBM groups")
        res = super().load_state_dict(state_dict, strict=strict)
        self.repack()
        return res

    def _apply(self, fn, *a, **k):  # parameters are views into flat HBM buffers: moving them breaks the engine
        raise RuntimeError("osrl_amd models are bound to their HIP device at construction; .to()/.cuda()/.cpu() "
                           "are unsupported (pass device= to the constructor)")

    def setup_optimizers(self, actor_lr, critic_lr, alpha_lr, vae_lr):
        """cpq.py:232-238 -- Adam(lr) x4; the state lives in the flat groups (m, v) on device."""
        self._lrs = dict(actor=actor_lr, critic=critic_lr, cost_critic=critic_lr, vae=vae_lr)
        self.alpha_lr = alpha_lr

    def engine(self, batch_size: int, **kw):
        from ..common.checkpoint import engine_handoff
        from ..engine.cpq import CPQEngine
        if self._engine is None or self._engine.B != batch_size or kw:
            if self._lrs is None:
                raise RuntimeError("call setup_optimizers() (or build a CPQTrainer) before training")
            old, self._engine = self._engine, CPQEngine(self, batch_size, **kw)
            engine_handoff(self, self._engine, old)
        return self._engine

    def sync_weight(self):
        """cpq.py:224-230.  The Polyak update is fused into each group's optimizer kernel inside
        train_one_step (same result: no target is read between its group's step and the step end)."""
        return None

    def fast_policy(self):
        """The B = 1 latency path (engine/act.py): one kernel launch per ``act()``, pinned-memory I/O."""
        if self._fast is None:
            from ..common.net import actor_head_desc
            from ..engine.act import FastPolicy
            self._fast = FastPolicy("gauss", self.device, self.state_dim, self.action_dim, actor_head_desc(self.actor),
                                    max_action=self.max_action)
        return self._fast

    def act(self, obs: np.ndarray, deterministic: bool = False, with_logprob: bool = False, eps=None):
        """cpq.py:240-252: single observation -> (max_action * tanh(u), logp).  ``eps``: optional explicit noise."""
        fp = self._fast if self._fast is not None else self.fast_policy()
        return fp.act1(obs, deterministic) if eps is None else fp.act(obs, deterministic, noise=eps)


class CPQTrainer:
    """cpq.py:255-347.  ``stats_mode``: "sync" stores python floats in the logger every step like the
    reference (one host sync per step); "lazy" (default) stores ``LazyStat`` objects that read the
    device statistics ring only when converted to float, so the step loop never blocks."""

    def __init__(self, model: CPQ, env=None, logger=None, actor_lr: float = 1e-4,
                 critic_lr: float = 1e-4, alpha_lr: float = 1e-4, vae_lr: float = 1e-4,
                 reward_scale: float = 1.0, cost_scale: float = 1.0, device="cuda",
                 stats_mode: str = "lazy", use_graph: bool = True) -> None:
        self.model = model
        if logger is None:  # the reference default (a fresh one per trainer: no shared mutable default)
            from ..common.logger import DummyLogger
            logger = DummyLogger()
        self.logger = logger
        self.env = env
        self.reward_scale = reward_scale
        self.cost_scale = cost_scale
        self.device = device
        self.stats_mode = stats_mode
        self.use_graph = use_graph
        self.model.setup_optimizers(actor_lr, critic_lr, alpha_lr, vae_lr)

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done, noise=None):
        """One CPQ gradient step (vae -> critic -> cost critic -> actor -> Polyak), cpq.py:294-313.
        ``noise``: optional dict of explicit standard-normal tensors (parity tests);
        when omitted the noise is drawn on device (Philox) inside the step."""
        eng = self.model.engine(observations.shape[0])
        eng.step(observations, next_observations, actions, rewards, costs, done, noise=noise,
                 use_graph=self.use_graph and noise is None)
        from ..common.logger import store_stats
        store_stats(self.logger, eng.st, self.stats_mode)

    def evaluate(self, eval_episodes):
        """cpq.py:315-328.  With a ``VecSyntheticSafeEnv`` as ``self.env`` the episodes run as one batch on device
        (engine/rollout.py); any other (gym-style) env takes the reference's episode-by-episode loop."""
        from ..common.synthetic_env import VecSyntheticSafeEnv
        if isinstance(self.env, VecSyntheticSafeEnv):
            from ..engine.rollout import evaluate_batched
            r, c, n = evaluate_batched(self, "cpq", eval_episodes, self.cost_scale)
            return r / self.reward_scale, c / self.cost_scale, n
        self.model.eval()
        rets, costs, lens = [], [], []
        for _ in range(eval_episodes):
            r, l, c = self.rollout()
            rets.append(r)
            lens.append(l)
            costs.append(c)
        self.model.train()
        return np.mean(rets) / self.reward_scale, np.mean(costs) / self.cost_scale, np.mean(lens)

    @torch.no_grad()
    def rollout(self):
        """cpq.py:330-347."""
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
