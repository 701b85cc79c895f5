"""A deterministic, dependency-free stand-in for the DSRL gym environments (absent from every container):
contractive linear dynamics, quadratic reward, half-space indicator cost.  It exists so that
``Trainer.evaluate()/rollout()`` can run end to end and be compared against the CPU oracle driving the same
environment (the metric's "cost-return gap vs ref").  gymnasium-style API.

``SyntheticSafeEnv`` is the scalar host environment (numpy); ``VecSyntheticSafeEnv`` is E copies of it resident
in HBM and stepped by one HIP launch (csrc/env.hip) for the batched evaluate().  Episode ``e``
of the vector environment starts where ``SyntheticSafeEnv.reset(seed=base_seed + e)`` starts."""
from __future__ import annotations

import numpy as np


class SyntheticSafeEnv:
    COST_THRESHOLD = 0.75

    def __init__(self, state_dim: int, action_dim: int, episode_len: int = 50, seed: int = 0, max_action: float = 1.0,
                 init_noise: float = 0.0):
        """``init_noise`` > 0: ``reset(seed=k)`` starts at ``s0 + init_noise * N(0, I)`` drawn from
        ``RandomState(k)`` (distinct episodes); ``reset()`` without a seed always starts at ``s0``."""
        rs = np.random.RandomState(seed)
        self.init_noise = float(init_noise)
        A = rs.randn(state_dim, state_dim)
        self.A = (0.9 * A / np.abs(np.linalg.eigvals(A)).max()).astype(np.float32)
        self.Bm = (0.3 * rs.randn(state_dim, action_dim)).astype(np.float32)
        self.w = rs.randn(state_dim).astype(np.float32)
        self.goal = rs.randn(state_dim).astype(np.float32) * 0.5
        self.s0 = rs.randn(state_dim).astype(np.float32)
        self.episode_len, self.max_action = episode_len, max_action
        self.state_dim, self.action_dim = state_dim, action_dim
        self.t, self.s = 0, self.s0.copy()

    def initial_state(self, seed=None) -> np.ndarray:
        if seed is None or self.init_noise == 0.0:
            return self.s0.copy()
        d = np.random.RandomState(int(seed)).randn(self.state_dim).astype(np.float32)
        return (self.s0 + np.float32(self.init_noise) * d).astype(np.float32)

    def reset(self, seed=None):
        self.t, self.s = 0, self.initial_state(seed)
        return self.s.copy(), {"cost": 0.0}

    def step(self, action):
        a = np.clip(np.asarray(action, np.float32).reshape(-1), -self.max_action, self.max_action)
        self.s = (self.A @ self.s + self.Bm @ a).astype(np.float32)
        self.t += 1
        reward = float(1.0 - 0.1 * np.sum((self.s - self.goal) ** 2))
        cost = float(self.s @ self.w > self.COST_THRESHOLD)
        return self.s.copy(), reward, False, self.t >= self.episode_len, {"cost": cost}


class VecSyntheticSafeEnv:
    """E episodes of one ``SyntheticSafeEnv`` in HBM: ``state[E, od]`` plus the per-episode accumulators
    ``acc[E, 4] = (return, cost * cost_scale, length, done)``; ``step(actions, obs_out)`` is one HIP launch."""

    def __init__(self, env: SyntheticSafeEnv, episodes: int, device="cuda", base_seed: int = 0):
        import torch

        from .. import _lib as L
        from ..engine.core import require_cuda
        self.env, self.E, self.base_seed = env, int(episodes), int(base_seed)
        self.device = require_cuda(device)
        L.load()
        t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=self.device)  # noqa: E731
        self.At, self.Bt, self.w, self.goal = t(env.A.T), t(env.Bm.T), t(env.w), t(env.goal)
        # the initial states are a function of (env, base_seed) only: drawn once, kept in HBM, reset() is a device copy
        self.state0 = t(np.stack([env.initial_state(self.base_seed + e) for e in range(self.E)]))
        self.state = torch.zeros(self.E, env.state_dim, dtype=torch.float32, device=self.device)
        self.acc = torch.zeros(self.E, 4, dtype=torch.float32, device=self.device)
        self.state_dim, self.action_dim, self.episode_len = env.state_dim, env.action_dim, env.episode_len

    def desc(self, episode_len: int, cost_scale: float):
        from .. import _lib as L
        d = L.EnvT()
        d.At, d.Bt, d.w, d.goal = self.At.data_ptr(), self.Bt.data_ptr(), self.w.data_ptr(), self.goal.data_ptr()
        d.state_dim, d.action_dim, d.episode_len = self.state_dim, self.action_dim, int(episode_len)
        d.max_action, d.cost_threshold, d.cost_scale = self.env.max_action, self.env.COST_THRESHOLD, cost_scale
        return d

    def reset(self, obs_out) -> None:
        """Initial states of episodes base_seed .. base_seed+E-1 -> ``state`` and ``obs_out[:, :od]``; zero totals."""
        self.state.copy_(self.state0)
        obs_out[:, :self.state_dim].copy_(self.state0)
        self.acc.zero_()

    def step(self, desc, actions, obs_out, step_out=None) -> None:
        """``step_out`` (optional [E,2]) receives this step's (reward, raw cost) -- CDT's to-go bookkeeping."""
        from .. import _lib as L
        from ..engine.core import cur_stream
        L.check(L.load().osrl_env_step(desc, actions.data_ptr(), self.state.data_ptr(), obs_out.data_ptr(),
                                       obs_out.stride(0), self.acc.data_ptr(),
                                       None if step_out is None else step_out.data_ptr(), self.E, cur_stream()),
                "osrl_env_step")
