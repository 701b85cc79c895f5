"""CPU baseline for bench.py: the CPQ train step restated with torch CPU tensors + autograd + torch.optim.Adam.

TEST INFRASTRUCTURE, like everything under oracle/: only tests/, ``__graft_entry__.smoke()`` and bench.py's
``cpu_baseline`` leg may import it; nothing under osrl_amd/ does.  It exists because the reference's own CPU path IS
torch-on-CPU (aten GEMMs + autograd + torch.optim.Adam, ``torch.set_num_threads(4)`` in examples/train/train_cpq.py:35),
so this is the closer stand-in for "the reference timed on this host" than the numpy port: same library, same
threading runtime, the same kind of per-op dispatch.  It is an independent restatement of

  CPQ.vae_loss / critic_loss / cost_critic_loss / actor_loss / sync_weight   osrl/algorithms/cpq.py:125-230
  CPQTrainer.train_one_step                                                   osrl/algorithms/cpq.py:294-313
  SquashedGaussianMLPActor, EnsembleQCritic, VAE                              osrl/common/net.py:152-205,208-268,290-339

over a flat {state_dict key: tensor} parameter set (tests/cases.py ``make_params`` naming == the reference's
``state_dict()`` naming).  Pinned by tests/test_oracle_golden.py::test_torch_cpu_baseline_matches_numpy_oracle against
the numpy oracle, which itself is pinned to the reference's golden vectors.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0  # net.py:148-149


def _prefix_layers(p: Dict[str, torch.Tensor], prefix: str) -> List[str]:
    """Sorted ``prefix.<i>`` stems of an nn.Sequential's Linear layers (even indices; activations sit between)."""
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in p if k.startswith(prefix + ".") and
                  k[len(prefix) + 1:].split(".")[0].isdigit()})
    return [f"{prefix}.{i}" for i in idx]


class TorchCPQ:
    def __init__(self, params, *, max_action: float, sample_action_num: int = 10, gamma: float = 0.99,
                 tau: float = 0.005, beta: float = 0.5, qc_scalar: float = 1.5, cost_limit: float = 10,
                 episode_len: int = 300, actor_lr=1e-4, critic_lr=1e-3, alpha_lr=1e-4, vae_lr=1e-3):
        self.p = {k: torch.tensor(v, dtype=torch.float32) for k, v in params.items()}
        p = self.p
        self.max_action, self.N, self.gamma, self.tau, self.beta = max_action, sample_action_num, gamma, tau, beta
        self.q_thres = cost_limit * (1 - gamma ** episode_len) / (1 - gamma) / episode_len  # cpq.py:102-105
        self.qc_thres = qc_scalar * self.q_thres
        self.log_alpha = torch.tensor(0.0)  # cpq.py:93 (updated by hand, cpq.py:193-195)
        self.alpha_lr = alpha_lr
        nets = lambda name: sorted({k.split(".")[2] for k in p if k.startswith(name + ".q_nets.")}, key=int)  # noqa: E731
        self.q = [f"critic.q_nets.{i}" for i in nets("critic")]
        self.qc = [f"cost_critic.q_nets.{i}" for i in nets("cost_critic")]
        self.q_old = [f"critic_old.q_nets.{i}" for i in nets("critic_old")]
        self.qc_old = [f"cost_critic_old.q_nets.{i}" for i in nets("cost_critic_old")]
        grp = lambda pre: [p[k] for k in p if k.startswith(pre + ".")]  # noqa: E731
        for pre in ("actor", "critic", "cost_critic", "vae"):
            for t in grp(pre):
                t.requires_grad_(True)
        self.opt_actor = torch.optim.Adam(grp("actor"), lr=actor_lr)
        self.opt_critic = torch.optim.Adam(grp("critic"), lr=critic_lr)
        self.opt_cost = torch.optim.Adam(grp("cost_critic"), lr=critic_lr)
        self.opt_vae = torch.optim.Adam(grp("vae"), lr=vae_lr)

    # ---- modules (net.py) as functions of the flat parameter dict
    def _seq(self, prefix: str, x, hidden_act, out_act=None):
        stems = _prefix_layers(self.p, prefix)
        for i, s in enumerate(stems):
            x = F.linear(x, self.p[s + ".weight"], self.p[s + ".bias"])
            if i + 1 < len(stems):
                x = hidden_act(x)
            elif out_act is not None:
                x = out_act(x)
        return x

    def _q_all(self, prefixes, obs, act):  # EnsembleQCritic.forward, net.py:229-233
        x = torch.cat([obs, act], 1)
        return [self._seq(pre, x, F.relu).squeeze(-1) for pre in prefixes]

    def _actor_dist(self, obs):  # net.py:176-181
        h = self._seq("actor.net", obs, F.relu, F.relu)
        mu = F.linear(h, self.p["actor.mu_layer.weight"], self.p["actor.mu_layer.bias"])
        ls = F.linear(h, self.p["actor.log_std_layer.weight"], self.p["actor.log_std_layer.bias"])
        return mu, torch.exp(torch.clamp(ls, LOG_STD_MIN, LOG_STD_MAX))

    def _actor(self, obs, eps):  # cpq.py:115-123 (the log-prob is not used by any CPQ loss)
        mu, std = self._actor_dist(obs)
        return torch.tanh(mu + std * eps) * self.max_action

    def _vae_encode(self, obs, act):  # net.py:319-326
        h = F.relu(F.linear(torch.cat([obs, act], 1), self.p["vae.e1.weight"], self.p["vae.e1.bias"]))
        h = F.relu(F.linear(h, self.p["vae.e2.weight"], self.p["vae.e2.bias"]))
        mean = F.linear(h, self.p["vae.mean.weight"], self.p["vae.mean.bias"])
        ls = torch.clamp(F.linear(h, self.p["vae.log_std.weight"], self.p["vae.log_std.bias"]), -4, 15)
        return mean, torch.exp(ls)

    def _vae_decode(self, obs, z):  # net.py:332-339
        h = F.relu(F.linear(torch.cat([obs, z], 1), self.p["vae.d1.weight"], self.p["vae.d1.bias"]))
        h = F.relu(F.linear(h, self.p["vae.d2.weight"], self.p["vae.d2.bias"]))
        return self.max_action * torch.tanh(F.linear(h, self.p["vae.d3.weight"], self.p["vae.d3.bias"]))

    @staticmethod
    def _kl(mean, std):  # cpq.py:128,181
        return -0.5 * (1 + torch.log(std.pow(2)) - mean.pow(2) - std.pow(2))

    def _polyak(self, tgt: str, src: str):  # cpq.py:107-113
        with torch.no_grad():
            for k, v in self.p.items():
                if k.startswith(tgt + "."):
                    v.mul_(1 - self.tau).add_(self.p[src + k[len(tgt):]], alpha=self.tau)

    # ---- one gradient step
    def train_one_step(self, observations, next_observations, actions, rewards, costs, done, noise) -> Dict[str, float]:
        t = lambda a: torch.as_tensor(a, dtype=torch.float32)  # noqa: E731
        obs, nobs, act, rew, cost, done = (t(a) for a in (observations, next_observations, actions, rewards, costs,
                                                           done))
        nz = {k: t(v) for k, v in noise.items()}
        N, g = self.N, self.gamma
        B = obs.shape[0]
        stats = {}

        # vae_loss  cpq.py:125-135
        mean, std = self._vae_encode(obs, act)
        recon = self._vae_decode(obs, mean + std * nz["eps_vae"])
        loss_vae = F.mse_loss(recon, act) + self.beta * self._kl(mean, std).mean()
        self.opt_vae.zero_grad()
        loss_vae.backward()
        self.opt_vae.step()
        stats["loss/loss_vae"] = loss_vae.item()

        # critic_loss  cpq.py:137-153
        with torch.no_grad():
            na = self._actor(nobs, nz["eps_next_c"])
            q_t = torch.stack(self._q_all(self.q_old, nobs, na)).min(0).values
            qc_t = torch.stack(self._q_all(self.qc_old, nobs, na)).min(0).values
            backup = rew + g * (1 - done) * (qc_t <= self.q_thres) * q_t
        loss_c = sum(F.mse_loss(q, backup) for q in self._q_all(self.q, obs, act))
        self.opt_critic.zero_grad()
        loss_c.backward()
        self.opt_critic.step()
        stats["loss/critic_loss"] = loss_c.item()

        # cost_critic_loss  cpq.py:155-201
        with torch.no_grad():
            na = self._actor(nobs, nz["eps_next_cc"])
            backup = cost + g * torch.stack(self._q_all(self.qc_old, nobs, na)).min(0).values
            mu, std_a = self._actor_dist(obs)
            sampled = (mu[None] + std_a[None] * nz["eps_ood"]).reshape(N * B, -1)  # pre-tanh, cpq.py:166
            stacked = obs[None].expand(N, B, obs.shape[1]).reshape(N * B, -1)
            qc_s = torch.stack(self._q_all(self.qc_old, stacked, sampled)).min(0).values.reshape(N, B)
            m_o, s_o = self._vae_encode(stacked, sampled)
            kl = self._kl(m_o, s_o).mean(1).reshape(N, B)
            quant = torch.quantile(kl, 0.75)
            qc_ood = ((kl >= quant) * qc_s).mean(0)
        loss_cc = sum(F.mse_loss(qc, backup) for qc in self._q_all(self.qc, obs, act)) \
            - self.log_alpha.exp() * (qc_ood.mean() - self.qc_thres)
        self.opt_cost.zero_grad()
        loss_cc.backward()
        self.opt_cost.step()
        with torch.no_grad():  # cpq.py:193-195
            self.log_alpha += self.alpha_lr * self.log_alpha.exp() * (self.qc_thres - qc_ood.mean())
            self.log_alpha.clamp_(-5.0, 5.0)
        stats["loss/cost_critic_loss"] = loss_cc.item()
        stats["loss/alpha_value"] = math.exp(self.log_alpha.item())

        # actor_loss  cpq.py:203-222 (critics frozen: their .grad is discarded by the next zero_grad)
        a = self._actor(obs, nz["eps_actor"])
        q_pi = torch.stack(self._q_all(self.q, obs, a)).min(0).values
        with torch.no_grad():
            qc_pi = torch.stack(self._q_all(self.qc, obs, a)).min(0).values
        loss_a = -((qc_pi <= self.q_thres) * q_pi).mean()
        self.opt_actor.zero_grad()
        loss_a.backward()
        self.opt_actor.step()
        stats["loss/actor_loss"] = loss_a.item()

        # sync_weight  cpq.py:224-230
        self._polyak("critic_old", "critic")
        self._polyak("cost_critic_old", "cost_critic")
        self._polyak("actor_old", "actor")
        return stats
