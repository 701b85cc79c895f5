"""Behaviour cloning on MI355X behind the reference's API (osrl/algorithms/bc.py)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from ..common.logger import store_stats
from ..common.net import MLPActor, bind_group, plan_group
from ..engine.core import FlatGroup, require_cuda


class BC(nn.Module):
    """bc.py:12-64."""

    def __init__(self, state_dim: int, action_dim: int, max_action: float, a_hidden_sizes: list = [128, 128],
                 episode_len: int = 300, device: str = "cuda"):
        super().__init__()
        self.state_dim, self.action_dim, self.max_action = state_dim, action_dim, max_action
        self.a_hidden_sizes = list(a_hidden_sizes)
        self.episode_len = episode_len
        self.device = str(device)
        dev = require_cuda(device)
        self.actor = MLPActor(state_dim, action_dim, self.a_hidden_sizes, nn.ReLU, max_action)
        g = FlatGroup("actor", dev)
        plan_group(g, "actor", self.actor)
        g.finalize()
        bind_group(g, "actor", self.actor)
        self.groups: Dict[str, FlatGroup] = {"actor": g}
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

    def setup_optimizers(self, actor_lr):
        self._lrs = dict(actor=actor_lr)

    def engine(self, batch_size: int, **kw):
        from ..common.checkpoint import engine_handoff
        from ..engine.bc import BCEngine
        if self._engine is None or self._engine.B != batch_size or kw:
            if self._lrs is None:
                raise RuntimeError("call setup_optimizers() (or build a BCTrainer) before training")
            old, self._engine = self._engine, BCEngine(self, batch_size, **kw)
            engine_handoff(self, self._engine, old)
        return self._engine

    @torch.no_grad()
    def fast_policy(self):
        """The B = 1 latency path (engine/act.py): one kernel launch per ``act()``, pinned-memory I/O."""
        if getattr(self, "_fast", None) is None:
            from ..common.net import net_desc_seq
            from ..engine.act import FastPolicy
            self._fast = FastPolicy("mlp", self.device, self.actor.pi[0].in_features, self.action_dim,
                                    net_desc_seq([self.actor.pi], float(self.actor.act_limit)))
        return self._fast

    def act(self, obs):
        """bc.py:57-64: single observation -> action (numpy)."""
        return self.fast_policy().act(obs)[0]


class BCTrainer:
    """bc.py:67-145."""

    def __init__(self, model: BC, env=None, logger=None, actor_lr: float = 1e-4, bc_mode: str = "all",
                 cost_limit: int = 10, device="cuda", stats_mode: str = "lazy", use_graph: bool = True):
        self.model, self.logger, self.env, self.device = model, logger, env, device
        self.bc_mode, self.cost_limit = bc_mode, cost_limit
        self.stats_mode, self.use_graph = stats_mode, use_graph
        self.model.setup_optimizers(actor_lr)

    def set_target_cost(self, target_cost):
        self.cost_limit = target_cost

    def train_one_step(self, observations, actions):
        eng = self.model.engine(observations.shape[0])
        eng.step(observations, actions, use_graph=self.use_graph)
        store_stats(self.logger, eng.st, self.stats_mode)

    def evaluate(self, eval_episodes):
        """bc.py:111-123.  A ``VecSyntheticSafeEnv`` as ``self.env`` runs the episodes as one batch on device."""
        from ..common.synthetic_env import VecSyntheticSafeEnv
        if getattr(self.model, "_engine", None) is not None:
            self.model._engine.check_health()  # never evaluate parameters a failed one-launch step left behind
        if isinstance(self.env, VecSyntheticSafeEnv):
            from ..engine.rollout import evaluate_batched
            extra = float(self.cost_limit) if self.bc_mode == "multi-task" else None
            return evaluate_batched(self, "bc", eval_episodes, 1.0, extra)
        self.model.eval()
        rets, costs, lens = [], [], []
        for _ in range(eval_episodes):
            r, l, c = self.rollout()
            rets.append(r); lens.append(l); costs.append(c)
        self.model.train()
        return np.mean(rets), np.mean(costs), np.mean(lens)  # bc.py:123 does not rescale

    @torch.no_grad()
    def rollout(self):
        ep_ret, ep_cost, ep_len = 0.0, 0.0, 0
        obs, info = self.env.reset()
        if self.bc_mode == "multi-task":
            obs = np.append(obs, self.cost_limit)
        for _ in range(self.model.episode_len):
            act = self.model.act(obs)
            obs_next, reward, terminated, truncated, info = self.env.step(act)
            if self.bc_mode == "multi-task":
                obs_next = np.append(obs_next, self.cost_limit)
            obs = obs_next
            ep_ret += reward
            ep_len += 1
            ep_cost += info["cost"]
            if terminated or truncated:
                break
        return ep_ret, ep_len, ep_cost
