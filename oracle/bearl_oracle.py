"""CPU oracle for BEAR-Lagrangian ``train_one_step`` (osrl/algorithms/bearl.py of the reference).

TEST INFRASTRUCTURE ONLY (same rules as osrl_oracle.py): nothing under ``osrl_amd/`` may import it.
numpy restatement with hand-derived backward passes, built from the blocks of ``osrl_oracle`` (MLP, Adam,
SquashedGaussianActor, VAE, q_forward, PID); every method cites the reference file:line it follows.
PINNED by ``tests/golden/bearl_*.npz`` -- captured by importing the reference (``tests/golden/make_golden.py``).

Noise (SURVEY.md 8a-RNG style, the reference's draw order within one step):
  eps_vae [B, 2ad]      VAE reparametrisation                       bearl.py:143 -> net.py:327
  eps_c   [N*B, ad]     actor_old rsample on repeat_interleave(obs') bearl.py:161
  eps_cc  [N*B, ad]     same, cost critic                            bearl.py:188
  z_mmd   [B, M, 2ad]   decode_multiple's latent draw (clamped here) bearl.py:221 -> net.py:343-346
  eps_pi  [B*M, ad]     actor rsample on the stacked observations    bearl.py:228
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .osrl_oracle import (PID, VAE, Adam, Array, SquashedGaussianActor, State, _keys_with_prefix, _q_prefixes,
                          _vae_keys, q_forward, soft_update)


def mmd_and_grad(x: Array, y: Array, sigma: float, kernel: str):
    """``mmd_loss_gaussian`` / ``mmd_loss_laplacian`` (bearl.py:277-312) for x, y [B, M, d]: per-row
    sqrt(mean k(x,x) + mean k(y,y) - 2 mean k(x,y) + 1e-6), and its gradient w.r.t. y (x carries none: the VAE is
    frozen, bearl.py:216-217)."""
    M = x.shape[1]

    def pair(a, b):
        d = a[:, :, None, :] - b[:, None, :, :]  # [B, M, M, d]: a_i - b_j
        if kernel == "gaussian":
            k = np.exp(-(d ** 2).sum(-1) / (2.0 * sigma))
            dk_db = k[..., None] * d / sigma            # d k_ij / d b_j
        else:
            k = np.exp(-np.abs(d).sum(-1) / (2.0 * sigma))
            dk_db = k[..., None] * np.sign(d) / (2.0 * sigma)
        return k, dk_db

    kxx, _ = pair(x, x)
    kxy, dxy = pair(x, y)
    kyy, dyy = pair(y, y)
    inner = kxx.mean((1, 2)) + kyy.mean((1, 2)) - 2.0 * kxy.mean((1, 2)) + 1e-6
    mmd = np.sqrt(inner)
    # d mean(kyy)/dy_j: y_j appears as b_j (column j) and as a_j (row j); the kernel is symmetric
    g_yy = 2.0 * dyy.sum(1) / (M * M)
    g_xy = dxy.sum(1) / (M * M)
    dy = (g_yy - 2.0 * g_xy) / (2.0 * mmd)[:, None, None]
    return mmd, dy


class OracleBEARL:
    """BEARL + BEARLTrainer.train_one_step (bearl.py:393-417)."""

    def __init__(self, params: State, *, max_action: float, sample_action_num: int = 10, gamma=0.99, tau=0.005,
                 beta=0.5, lmbda=0.75, mmd_sigma=50.0, target_mmd_thresh=0.05, num_samples_mmd_match=10,
                 PID_gains=(0.1, 0.003, 0.001), kernel="gaussian", cost_limit=10, episode_len=300,
                 start_update_policy_step=20_000, actor_lr=1e-3, critic_lr=1e-3, alpha_lr=1e-3, vae_lr=1e-3,
                 dtype=np.float32):
        self.p = {k: np.array(v, dtype=dtype) for k, v in params.items()}
        p = self.p
        self.dtype = dtype
        self.max_action, self.N, self.M = max_action, sample_action_num, num_samples_mmd_match
        self.gamma, self.tau, self.beta, self.lmbda = gamma, tau, beta, lmbda
        self.sigma, self.thresh, self.kernel = mmd_sigma, target_mmd_thresh, kernel
        self.start = start_update_policy_step
        self.alpha_lr = alpha_lr
        self.qc_thres = cost_limit * (1 - gamma ** episode_len) / (1 - gamma) / episode_len  # bearl.py:120-121
        self.controller = PID(*PID_gains, self.qc_thres)
        self.log_alpha = 0.0  # bearl.py:110
        self.n_train_steps = 0
        self.vae = VAE(max_action, "vae")
        self.actor = SquashedGaussianActor(p, "actor")
        self.actor_old = SquashedGaussianActor(p, "actor_old")
        self.nets = {n: _q_prefixes(p, n, "q1_nets") + _q_prefixes(p, n, "q2_nets")
                     for n in ("critic", "cost_critic", "critic_old", "cost_critic_old")}
        self.nq = {n: len(_q_prefixes(p, n, "q1_nets")) for n in self.nets}
        self.opt_actor = Adam(_keys_with_prefix(p, "actor"), actor_lr)
        self.opt_critic = Adam(_keys_with_prefix(p, "critic"), critic_lr)
        self.opt_cost = Adam(_keys_with_prefix(p, "cost_critic"), critic_lr)
        self.opt_vae = Adam(_vae_keys(), vae_lr)

    def act(self, obs: Array) -> Array:
        """BEARL.act deterministic (bearl.py:337-350): max_action * tanh(mu)."""
        return self.max_action * self.actor.forward(self.p, np.asarray(obs, self.dtype), None)["a"]

    def _targets(self, name_old, nobs, eps):
        """bearl.py:158-169: actor_old samples (NOT scaled by max_action) on the N-fold repeated next observations,
        lambda-weighted twin min/max, max over the N samples."""
        p, N = self.p, self.N
        B = nobs.shape[0]
        obs_n = np.repeat(nobs, N, 0)
        a = self.actor_old.forward(p, obs_n, eps)["a"]
        qs, _ = q_forward(p, self.nets[name_old], np.concatenate([obs_n, a], 1))
        n1 = self.nq[name_old]
        q1, q2 = np.min(np.stack(qs[:n1]), 0), np.min(np.stack(qs[n1:]), 0)
        q = self.lmbda * np.minimum(q1, q2) + (1 - self.lmbda) * np.maximum(q1, q2)
        return q.reshape(B, N).max(1)

    def _critic_update(self, name, x, backup, opt):
        B = x.shape[0]
        qs, caches = q_forward(self.p, self.nets[name], x)
        loss = sum(((q - backup) ** 2).mean() for q in qs)  # bearl.py:173-174
        grads: State = {}
        for q, (net, cache) in zip(qs, caches):
            net.backward(self.p, cache, (2 * (q - backup) / B)[:, None], grads, need_dx=False)
        opt.step(self.p, grads)
        return float(loss)

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done,
                       noise: Dict[str, Array]) -> Dict[str, float]:
        dt = self.dtype
        obs, nobs, act = (np.asarray(a, dt) for a in (observations, next_observations, actions))
        rew, cost, done = (np.asarray(a, dt) for a in (rewards, costs, done))
        nz = {k: np.asarray(v, dt) for k, v in noise.items()}
        p, g, M = self.p, self.gamma, self.M
        B, od = obs.shape
        ad = act.shape[1]
        stats: Dict[str, float] = {}

        loss_vae, gr = self.vae.loss_and_grads(p, obs, act, nz["eps_vae"], self.beta)  # bearl.py:142-153
        self.opt_vae.step(p, gr)
        stats["loss/loss_vae"] = float(loss_vae)

        x = np.concatenate([obs, act], 1)
        backup = rew + g * (1 - done) * self._targets("critic_old", nobs, nz["eps_c"])  # bearl.py:171
        stats["loss/critic_loss"] = self._critic_update("critic", x, backup.astype(dt), self.opt_critic)
        backup = cost + g * self._targets("cost_critic_old", nobs, nz["eps_cc"])  # bearl.py:198 (no done mask)
        stats["loss/cost_critic_loss"] = self._critic_update("cost_critic", x, backup.astype(dt), self.opt_cost)

        # ---- actor_loss  bearl.py:211-275
        obs_m = np.repeat(obs, M, 0)  # row b*M + j  (bearl.py:224-226; decode_multiple uses the same order)
        z = np.clip(nz["z_mmd"].reshape(B * M, -1), -0.5, 0.5)
        # decode_multiple returns (tanh(d3), d3): the raw, pre-tanh decoder output feeds the MMD (net.py:348-353)
        _, dcache = self.vae.dec.forward(p, np.concatenate([obs_m, z], 1))
        raw_vae = dcache[-2] @ p["vae.d3.weight"].T + p["vae.d3.bias"]
        fw = self.actor.forward(p, obs_m, nz["eps_pi"])
        u, a_s = fw["u"], fw["a"]
        mmd, dmmd_du = mmd_and_grad(raw_vae.reshape(B, M, ad), u.reshape(B, M, ad), self.sigma, self.kernel)
        a0 = a_s.reshape(B, M, ad)[:, 0, :]  # bearl.py:243-245: the critics see the first sample only
        xa = np.concatenate([obs, a0], 1)

        def minmin(name):
            qs, caches = q_forward(p, self.nets[name], xa)
            n1 = self.nq[name]
            s1, s2 = np.stack(qs[:n1]), np.stack(qs[n1:])
            i1, i2 = np.argmin(s1, 0), np.argmin(s2, 0)
            m1, m2 = s1[i1, np.arange(B)], s2[i2, np.arange(B)]
            w1 = np.where(m1 < m2, 1.0, np.where(m1 == m2, 0.5, 0.0)).astype(dt)
            sel = [w1 * (i1 == i) for i in range(n1)] + [(1 - w1) * (i2 == i) for i in range(len(qs) - n1)]
            return np.minimum(m1, m2), sel, caches

        q_val, sel_q, caches_q = minmin("critic")
        qc_val, sel_qc, caches_qc = minmin("cost_critic")
        mult = self.controller.control(qc_val)
        qc_penalty = ((qc_val - self.qc_thres) * mult).mean()
        alpha = float(np.exp(self.log_alpha))
        use_q = self.n_train_steps >= self.start  # bearl.py:254-259
        loss_a = ((-q_val if use_q else 0.0) + alpha * (mmd - self.thresh)).mean() + qc_penalty
        da0 = np.zeros((B, ad), dt)
        scratch: State = {}
        if use_q:
            for sel, (net, cache) in zip(sel_q, caches_q):
                da0 += net.backward(p, cache, (-sel / B)[:, None].astype(dt), scratch, True)[:, od:]
        for sel, (net, cache) in zip(sel_qc, caches_qc):
            da0 += net.backward(p, cache, (sel * mult / B)[:, None].astype(dt), scratch, True)[:, od:]
        du = (alpha / B) * dmmd_du  # [B, M, ad]
        du[:, 0, :] += da0 * (1 - a0 ** 2)  # tanh'
        gr = {}
        self.actor.backward(p, fw, du.reshape(B * M, ad).astype(dt), nz["eps_pi"], gr)
        self.opt_actor.step(p, gr)
        # bearl.py:265-268
        self.log_alpha = float(np.clip(self.log_alpha + self.alpha_lr * alpha * (mmd - self.thresh).mean(), -5.0, 5.0))
        self.n_train_steps += 1
        stats["loss/actor_loss"] = float(loss_a)
        stats["loss/mmd_loss"] = float(mmd.mean())
        stats["loss/qc_penalty"] = float(qc_penalty)
        stats["loss/lagrangian"] = float(mult)
        stats["loss/alpha_value"] = float(np.exp(self.log_alpha))

        soft_update(p, "critic_old", "critic", self.tau)  # bearl.py:329-335
        soft_update(p, "cost_critic_old", "cost_critic", self.tau)
        soft_update(p, "actor_old", "actor", self.tau)
        return stats
