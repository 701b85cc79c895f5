"""A deterministic, dependency-free stand-in for the DSRL gym environments (absent from every container):
contractive linear dynamics, quadratic reward, half-space indicator cost.  It exists so that
``Trainer.evaluate()/rollout()`` can run end to end and be compared against the CPU oracle driving the same
environment (the metric's "cost-return gap vs ref").  gymnasium-style API."""
from __future__ import annotations

import numpy as np


class SyntheticSafeEnv:
    def __init__(self, state_dim: int, action_dim: int, episode_len: int = 50, seed: int = 0, max_action: float = 1.0):
        rs = np.random.RandomState(seed)
        A = rs.randn(state_dim, state_dim)
        self.A = (0.9 * A / np.abs(np.linalg.eigvals(A)).max()).astype(np.float32)
        self.Bm = (0.3 * rs.randn(state_dim, action_dim)).astype(np.float32)
        self.w = rs.randn(state_dim).astype(np.float32)
        self.goal = rs.randn(state_dim).astype(np.float32) * 0.5
        self.s0 = rs.randn(state_dim).astype(np.float32)
        self.episode_len, self.max_action = episode_len, max_action
        self.state_dim, self.action_dim = state_dim, action_dim
        self.t, self.s = 0, self.s0.copy()

    def reset(self, seed=None):
        self.t, self.s = 0, self.s0.copy()
        return self.s.copy(), {"cost": 0.0}

    def step(self, action):
        a = np.clip(np.asarray(action, np.float32).reshape(-1), -self.max_action, self.max_action)
        self.s = (self.A @ self.s + self.Bm @ a).astype(np.float32)
        self.t += 1
        reward = float(1.0 - 0.1 * np.sum((self.s - self.goal) ** 2))
        cost = float(self.s @ self.w > 0.75)
        return self.s.copy(), reward, False, self.t >= self.episode_len, {"cost": cost}
