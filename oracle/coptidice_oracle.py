"""CPU oracle for COptiDICE ``COptiDICETrainer.train_one_step`` (osrl/algorithms/coptidice.py of the reference).

TEST INFRASTRUCTURE ONLY (same rules as osrl_oracle.py): nothing under ``osrl_amd/`` may import it.
numpy restatement with hand-derived backward passes from the blocks of ``osrl_oracle``; every step cites the
reference file:line it follows.  PINNED by ``tests/golden/coptidice_*.npz`` -- captured by importing the reference
(``tests/golden/make_golden.py``).

Noise, in the reference's draw order within one step: ``obs_eps [B, od]`` (coptidice.py:204), ``act_eps [B, ad]``
(:205), then the actor's own rsample [B, ad] inside ``forward(deterministic=False)`` (:207) whose sample is
discarded (only the distribution is used) -- not an input here.  ``tau`` / ``lmbda`` are scalar leaves with their own Adam optimizers (:96-97, :241-242); they are not in
``state_dict``.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from .osrl_oracle import LOG_STD_MAX, LOG_STD_MIN, Adam, Array, SquashedGaussianActor, State, _keys_with_prefix, \
    _q_prefixes, q_forward


def _softplus(x: float) -> float:
    return float(np.logaddexp(0.0, x))


def _sigmoid(x: float) -> float:
    return 1.0 / (1.0 + math.exp(-x))


def f_div(f_type: str):
    """get_f_div_fn (coptidice.py:15-38): returns (f, f', f'^{-1}, (f'^{-1})')."""
    if f_type == "chi2":
        return (lambda w: 0.5 * (w - 1) ** 2, lambda w: w - 1, lambda x: x + 1, lambda x: np.ones_like(x))
    if f_type == "softchi":
        f = lambda w: np.where(w < 1, w * (np.log(w + 1e-10) - 1) + 1, 0.5 * (w - 1) ** 2)  # noqa: E731
        fp = lambda w: np.where(w < 1, np.log(w + 1e-10) + w / (w + 1e-10) - 1, w - 1)  # noqa: E731
        g = lambda x: np.where(x < 0, np.exp(np.minimum(x, 0.0)), x + 1)  # noqa: E731
        gp = lambda x: np.where(x < 0, np.exp(np.minimum(x, 0.0)), 1.0)  # noqa: E731
        return f, fp, g, gp
    if f_type == "kl":
        return (lambda w: w * np.log(w + 1e-10), lambda w: np.log(w + 1e-10) + w / (w + 1e-10),
                lambda x: np.exp(x - 1), lambda x: np.exp(x - 1))
    raise NotImplementedError(f_type)


class ScalarAdam:
    """torch.optim.Adam on a single scalar leaf (coptidice.py:241-242)."""

    def __init__(self, lr):
        self.lr, self.t, self.m, self.v = lr, 0, 0.0, 0.0

    def step(self, p: float, g: float) -> float:
        self.t += 1
        self.m = 0.9 * self.m + 0.1 * g
        self.v = 0.999 * self.v + 0.001 * g * g
        return p - self.lr / (1 - 0.9 ** self.t) * self.m / (math.sqrt(self.v) / math.sqrt(1 - 0.999 ** self.t) + 1e-8)


class OracleCOptiDICE:
    def __init__(self, params: State, *, max_action: float, f_type: str, init_state_propotion: float,
                 observations_std, actions_std, gamma=0.99, alpha=0.5, cost_ub_epsilon=0.01, cost_limit=10,
                 episode_len=300, actor_lr=1e-3, critic_lr=1e-3, scalar_lr=1e-3, dtype=np.float32):
        self.p = {k: np.array(v, dtype=dtype) for k, v in params.items()}
        p = self.p
        self.dtype = dtype
        self.max_action, self.gamma, self.alpha, self.eps_ub = max_action, gamma, alpha, cost_ub_epsilon
        self.p0 = float(init_state_propotion)
        self.obs_std, self.act_std = np.asarray(observations_std, dtype), np.asarray(actions_std, dtype)
        self.qc_thres = cost_limit * (1 - gamma ** episode_len) / (1 - gamma) / episode_len  # coptidice.py:94-95
        self.tau, self.lmbda = 1.0, 1.0  # coptidice.py:96-97
        self.f, self.fp, self.g, self.gp = f_div(f_type)
        self.actor = SquashedGaussianActor(p, "actor")
        self.nu_nets = _q_prefixes(p, "nu_network", "q_nets")
        self.chi_nets = _q_prefixes(p, "chi_network", "q_nets")
        self.opt_actor = Adam(_keys_with_prefix(p, "actor"), actor_lr)
        self.opt_nu = Adam(_keys_with_prefix(p, "nu_network"), critic_lr)
        self.opt_chi = Adam(_keys_with_prefix(p, "chi_network"), critic_lr)
        self.opt_tau, self.opt_lmbda = ScalarAdam(scalar_lr), ScalarAdam(scalar_lr)

    def act(self, obs: Array) -> Array:
        """COptiDICE.act deterministic (coptidice.py:244-256): tanh(mu), no max_action scaling."""
        return self.actor.forward(self.p, np.asarray(obs, self.dtype), None)["a"]

    def _min_net(self, nets, x):
        """EnsembleQCritic.predict (net.py:236-238): min over the nets; returns value, argmin, caches."""
        qs, caches = q_forward(self.p, nets, x)
        s = np.stack(qs)
        i = np.argmin(s, 0)
        return s[i, np.arange(x.shape[0])], i, caches

    def _backward_min(self, nets_caches, idx, dval, grads):
        for k, (net, cache) in enumerate(nets_caches):
            net.backward(self.p, cache, (dval * (idx == k))[:, None].astype(self.dtype), grads, need_dx=False)

    def _optimal_w(self, obs, nobs, rew, cost, done, lam):
        """coptidice.py:122-131."""
        nu_s, i_s, c_s = self._min_net(self.nu_nets, obs)
        nu_n, i_n, c_n = self._min_net(self.nu_nets, nobs)
        e = rew - lam * cost + self.gamma * (1.0 - done) * nu_n - nu_s
        w = np.maximum(self.g(e / self.alpha), 0.0)
        return nu_s, nu_n, e, w, (i_s, c_s, i_n, c_n)

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done, is_init,
                       noise: Dict[str, Array]) -> Dict[str, float]:
        dt = self.dtype
        obs, nobs, act = (np.asarray(a, dt) for a in (observations, next_observations, actions))
        rew, cost, done, init = (np.asarray(a, dt) for a in (rewards, costs, done, is_init))
        p, gam, al, B = self.p, self.gamma, self.alpha, obs.shape[0]
        stats: Dict[str, float] = {}
        lam = _softplus(self.lmbda)  # coptidice.py:138
        nu_s, nu_n, e, w, (i_s, c_s, i_n, c_n) = self._optimal_w(obs, nobs, rew, cost, done, lam)
        Df = self.f(w).mean()

        tau_p = _softplus(self.tau)
        if self.eps_ub == 0:  # coptidice.py:150-155
            weighted_c = (w * cost).mean()
            chi_loss = tau_loss = D_kl = 0.0
        else:  # coptidice.py:157-185
            chi_s, j_s, d_s = self._min_net(self.chi_nets, obs)
            chi_n, j_n, d_n = self._min_net(self.chi_nets, nobs)
            ell = (1 - gam) * chi_s * init / self.p0 + w * (cost + gam * (1 - done) * chi_n - chi_s)
            logits = ell / tau_p
            z = logits - logits.max()
            lse = np.log(np.exp(z).sum())
            sm, lsm = np.exp(z - lse), z - lse
            weights, log_weights = sm * B, lsm + np.log(B)
            D_kl = (weights * log_weights - weights + 1).mean()
            weighted_c = (weights * w * cost).mean()
            chi_loss = (weights * ell).mean()
            # `weights` is NOT detached in the reference: d chi_loss / d ell_i = s_i * (1 + (ell_i - chi_loss)/tau')
            dell = sm * (1 + (ell - chi_loss) / tau_p)
            gr: State = {}
            self._backward_min(d_s, j_s, dell * ((1 - gam) * init / self.p0 - w), gr)
            gr_n: State = {}
            self._backward_min(d_n, j_n, dell * w * gam * (1 - done), gr_n)
            for k in gr_n:
                gr[k] = gr.get(k, 0) + gr_n[k]
            self.opt_chi.step(p, gr)
            tau_loss = tau_p * (self.eps_ub - D_kl)
            self.tau = self.opt_tau.step(self.tau, _sigmoid(self.tau) * (self.eps_ub - D_kl))

        # 1.2 nu loss  coptidice.py:188-194
        nu_loss = (1 - gam) * (nu_s * init / self.p0).mean() + (w * e - al * self.f(w)).mean()
        td_error = (e ** 2).mean()
        x = e / al
        dw_de = (self.g(x) > 0) * self.gp(x) / al
        de = (w + (e - al * self.fp(w)) * dw_de) / B
        gr = {}
        self._backward_min(c_s, i_s, (1 - gam) * init / (self.p0 * B) - de, gr)
        gr_n = {}
        self._backward_min(c_n, i_n, de * gam * (1 - done), gr_n)
        for k in gr_n:
            gr[k] = gr.get(k, 0) + gr_n[k]
        self.opt_nu.step(p, gr)

        # 1.3 lambda loss  coptidice.py:197-201
        lmbda_loss = lam * (self.qc_thres - weighted_c)
        self.lmbda = self.opt_lmbda.step(self.lmbda, _sigmoid(self.lmbda) * (self.qc_thres - weighted_c))

        # 2. policy extraction  coptidice.py:204-217 (updated nu network, the lambda' of the top of the step)
        obs_n = obs + noise["obs_eps"].astype(dt) * self.obs_std * dt(0.1)
        act_n = act + noise["act_eps"].astype(dt) * self.act_std * dt(0.1)
        fw = self.actor.forward(p, obs_n, None)
        ls = np.clip(fw["ls_raw"], LOG_STD_MIN, LOG_STD_MAX)
        mu, std = fw["mu"], fw["std"]
        logp = (-((act_n - mu) ** 2) / (2 * std * std) - ls - 0.5 * math.log(2 * math.pi)).sum(-1)
        _, _, _, w2, _ = self._optimal_w(obs, nobs, rew, cost, done, lam)
        actor_loss = -(w2 * logp).mean()
        coef = (-w2 / B)[:, None]
        dmu = coef * (act_n - mu) / (std * std)
        dls = coef * (((act_n - mu) ** 2) / (std * std) - 1.0)
        dls_raw = dls * ((fw["ls_raw"] >= LOG_STD_MIN) & (fw["ls_raw"] <= LOG_STD_MAX))
        pre, h = "actor", fw["h"]
        gr = {pre + ".mu_layer.weight": dmu.T @ h, pre + ".mu_layer.bias": dmu.sum(0),
              pre + ".log_std_layer.weight": dls_raw.T @ h, pre + ".log_std_layer.bias": dls_raw.sum(0)}
        dh = dmu @ p[pre + ".mu_layer.weight"] + dls_raw @ p[pre + ".log_std_layer.weight"]
        self.actor.trunk.backward(p, fw["cache"], dh.astype(dt), gr, need_dx=False)
        self.opt_actor.step(p, gr)

        stats.update({"loss/chi_loss": float(chi_loss), "loss/tau_loss": float(tau_loss), "loss/D_kl": float(D_kl),
                      "loss/Df": float(Df), "loss/td_error": float(td_error), "loss/nu_loss": float(nu_loss),
                      "loss/lmbda_loss": float(lmbda_loss), "loss/actor_loss": float(actor_loss),
                      "loss/tau": float(tau_p), "loss/lmbda": float(lam)})
        return stats
