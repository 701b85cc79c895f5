"""BCQ-Lagrangian on MI355X behind the reference's API (osrl/algorithms/bcql.py)."""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from ..common.logger import store_stats
from ..common.net import (VAE, EnsembleDoubleQCritic, MLPGaussianPerturbationActor, bind_group, plan_group)
from ..engine.core import FlatGroup, require_cuda


class LagrangianPIDController:
    """net.py:356-387 -- the integrator state lives in a 2-float device tensor ``state`` =
    [error_old, error_integral]; the update itself runs inside the actor-loss kernel
    (csrc/glue.hip bcq_actor_loss_kernel) so there is no host round trip."""

    def __init__(self, KP, KI, KD, thres, state: torch.Tensor) -> None:
        self.KP, self.KI, self.KD, self.thres = KP, KI, KD, thres
        self.state = state

    @property
    def error_old(self) -> float:
        return float(self.state[0].item())

    @property
    def error_integral(self) -> float:
        return float(self.state[1].item())


class BCQL(nn.Module):
    """bcql.py:15-112."""

    def __init__(self, state_dim: int, action_dim: int, max_action: float, a_hidden_sizes: list = [128, 128],
                 c_hidden_sizes: list = [128, 128], vae_hidden_sizes: int = 64, sample_action_num: int = 10,
                 gamma: float = 0.99, tau: float = 0.005, phi: float = 0.05, lmbda: float = 0.75,
                 beta: float = 0.5, PID: list = [0.1, 0.003, 0.001], num_q: int = 1, num_qc: int = 1,
                 cost_limit: int = 10, episode_len: int = 300, device: str = "cuda"):
        super().__init__()
        self.state_dim, self.action_dim, self.max_action = state_dim, action_dim, max_action
        self.latent_dim = self.action_dim * 2
        self.a_hidden_sizes, self.c_hidden_sizes = list(a_hidden_sizes), list(c_hidden_sizes)
        self.vae_hidden_sizes = vae_hidden_sizes
        self.sample_action_num = sample_action_num
        self.gamma, self.tau, self.phi, self.lmbda, self.beta = gamma, tau, phi, lmbda, beta
        self.KP, self.KI, self.KD = PID
        self.num_q, self.num_qc = num_q, num_qc
        self.cost_limit, self.episode_len = cost_limit, episode_len
        self.device = str(device)
        dev = require_cuda(device)

        # creation order of bcql.py:86-100 (actor, critic, cost_critic, vae)
        self.actor = MLPGaussianPerturbationActor(state_dim, action_dim, self.a_hidden_sizes, nn.Tanh, phi,
                                                  max_action)
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

        self.qc_thres = cost_limit * (1 - self.gamma ** self.episode_len) / (1 - self.gamma) / self.episode_len
        self.pid_state = torch.zeros(2, dtype=torch.float32, device=dev)
        self.controller = LagrangianPIDController(self.KP, self.KI, self.KD, self.qc_thres, self.pid_state)
        self._engine = None
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

    def _apply(self, fn, *a, **k):
        raise RuntimeError("osrl_amd models are bound to their HIP device at construction (pass device=)")

    def setup_optimizers(self, actor_lr, critic_lr, vae_lr):
        """bcql.py:218-226."""
        self._lrs = dict(actor=actor_lr, critic=critic_lr, cost_critic=critic_lr, vae=vae_lr)

    def engine(self, batch_size: int, **kw):
        from ..common.checkpoint import engine_handoff
        from ..engine.bcql import BCQLEngine
        if self._engine is None or self._engine.B != batch_size or kw:
            if self._lrs is None:
                raise RuntimeError("call setup_optimizers() (or build a BCQLTrainer) before training")
            old, self._engine = self._engine, BCQLEngine(self, batch_size, **kw)
            engine_handoff(self, self._engine, old)
        return self._engine

    def sync_weight(self):
        """bcql.py:228-234 -- fused into the per-group optimizer kernels of train_one_step."""
        return None

    @torch.no_grad()
    def act(self, obs, deterministic=False, with_logprob=False, z=None):
        """bcql.py:236-243 (stochastic: decode draws z unless given)."""
        if getattr(self, "_fast", None) is None:
            from ..common.net import net_desc_seq, vae_dec_desc
            from ..engine.act import FastPolicy
            self._fast = FastPolicy("bcq", self.device, self.state_dim, self.action_dim, vae_dec_desc(self.vae),
                                    max_action=float(self.actor.act_limit), net1=net_desc_seq([self.actor.pi], 1.0),
                                    latent_dim=self.latent_dim, phi=float(self.actor.phi))
        zz = None if z is None else (z.detach().cpu().numpy() if torch.is_tensor(z) else np.asarray(z))
        return self._fast.act(obs, deterministic, noise=zz)[0], None


class BCQLTrainer:
    """bcql.py:246-340."""

    def __init__(self, model: BCQL, env=None, logger=None, actor_lr: float = 1e-4,
                 critic_lr: float = 1e-4, vae_lr: float = 1e-4, reward_scale: float = 1.0,
                 cost_scale: float = 1.0, device="cuda", stats_mode: str = "lazy", use_graph: bool = True):
        self.model, self.logger, self.env = model, logger, env
        self.reward_scale, self.cost_scale, self.device = reward_scale, cost_scale, device
        self.stats_mode, self.use_graph = stats_mode, use_graph
        self.model.setup_optimizers(actor_lr, critic_lr, vae_lr)

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done, noise=None):
        eng = self.model.engine(observations.shape[0])
        eng.step(observations, next_observations, actions, rewards, costs, done, noise=noise,
                 use_graph=self.use_graph and noise is None)
        store_stats(self.logger, eng.st, self.stats_mode)

    def evaluate(self, eval_episodes):
        """bcql.py:308-321.  A ``VecSyntheticSafeEnv`` as ``self.env`` runs the episodes as one batch on device
        (the decode noise z is then drawn per env step from the device Philox stream)."""
        from ..common.synthetic_env import VecSyntheticSafeEnv
        if isinstance(self.env, VecSyntheticSafeEnv):
            from ..engine.rollout import evaluate_batched
            r, c, n = evaluate_batched(self, "bcql", eval_episodes, self.cost_scale)
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
        obs, info = self.env.reset()
        ep_ret, ep_cost, ep_len = 0.0, 0.0, 0
        for _ in range(self.model.episode_len):
            act, _ = self.model.act(obs)
            obs_next, reward, terminated, truncated, info = self.env.step(act)
            cost = info["cost"] * self.cost_scale
            obs = obs_next
            ep_ret += reward
            ep_len += 1
            ep_cost += cost
            if terminated or truncated:
                break
        return ep_ret, ep_len, ep_cost
