"""CPU baselines for bench.py, configs C1 / C3 / C5: the BC, BCQ-Lag and CDT train steps restated with torch CPU tensors +
autograd + torch.optim ("the build's own plain-PyTorch restatement" timed on the GPU box's host cores).

TEST INFRASTRUCTURE, like everything under oracle/: only tests/, ``__graft_entry__.smoke()`` and bench.py's
``cpu_baseline`` legs may import it; nothing under osrl_amd/ does.  The CPQ sibling is oracle/torch_cpq_cpu.py (same
rationale: the reference's own CPU path IS aten GEMMs + autograd + torch.optim.Adam at ``torch.set_num_threads(4)``).
Independent restatements over a flat {state_dict key: tensor} parameter set (tests/cases.py naming == the reference's
``state_dict()`` naming) of

  BC.actor_loss + BCTrainer.train_one_step                                      osrl/algorithms/bc.py:45-52,103-109
  BCQL.vae_loss / critic_loss / cost_critic_loss / actor_loss / sync_weight     osrl/algorithms/bcql.py:122-234
  BCQLTrainer.train_one_step                                                    osrl/algorithms/bcql.py:283-306
  MLPActor, MLPGaussianPerturbationActor, EnsembleDoubleQCritic, VAE, PID       osrl/common/net.py:33-85,245-387
  CDT.forward + TransformerBlock + DiagGaussianActor                            osrl/algorithms/cdt.py:166-265, net.py:391-441,530-533
  CDTTrainer.train_one_step (AdamW, clip_grad_norm_, warm-up LR, temperature)   osrl/algorithms/cdt.py:321-418

Pinned by tests/test_oracle_golden.py::test_torch_cpu_baselines_match_reference_goldens DIRECTLY against the vectors
captured from the reference (bc_small, bcql_small, bcql_pid, cdt_small, cdt_det): logged statistics of every step and
the parameters after the last one.  CDT covers the reference's train configuration (time embedding, return + cost
tokens, optional cost transform, one-layer stochastic or deterministic head, no cost-feature variants, no prefix).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def _t(a):
    return torch.as_tensor(a, dtype=torch.float32)


def _stems(p: Dict[str, torch.Tensor], prefix: str) -> List[str]:
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in p
                  if k.startswith(prefix + ".") and k[len(prefix) + 1:].split(".")[0].isdigit()})
    return [f"{prefix}.{i}" for i in idx]


def _mlp(p, prefix, x, hidden_act, out_act=None):
    stems = _stems(p, prefix)
    for i, s in enumerate(stems):
        x = F.linear(x, p[s + ".weight"], p[s + ".bias"])
        if i + 1 < len(stems):
            x = hidden_act(x)
        elif out_act is not None:
            x = out_act(x)
    return x


def _leafs(p, prefix):
    out = [v for k, v in p.items() if k.startswith(prefix + ".")]
    for v in out:
        v.requires_grad_(True)
    return out


# --------------------------------------------------------------------------------------------------------------------
class TorchBC:
    """bc.py:45-52,103-109 (bc_mode 'all' / 'safe' ...: plain MSE regression of the tanh-squashed MLP policy)."""

    def __init__(self, params, max_action: float, actor_lr: float = 1e-3):
        self.p = {k: torch.tensor(v, dtype=torch.float32) for k, v in params.items()}
        self.max_action = max_action
        self.opt = torch.optim.Adam(_leafs(self.p, "actor"), lr=actor_lr)

    def train_one_step(self, observations, actions) -> Dict[str, float]:
        pred = self.max_action * _mlp(self.p, "actor.pi", _t(observations), F.relu, torch.tanh)  # net.py:77-85
        loss = F.mse_loss(pred, _t(actions))
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return {"loss/actor_loss": loss.item()}


# --------------------------------------------------------------------------------------------------------------------
class TorchBCQL:
    def __init__(self, params, *, max_action: float, sample_action_num: int = 10, gamma=0.99, tau=0.005, phi=0.05,
                 lmbda=0.75, beta=0.5, PID_gains=(0.1, 0.003, 0.001), cost_limit=10, episode_len=300, actor_lr=1e-3,
                 critic_lr=1e-3, vae_lr=1e-3):
        self.p = {k: torch.tensor(v, dtype=torch.float32) for k, v in params.items()}
        p = self.p
        self.max_action, self.N, self.gamma, self.tau = max_action, sample_action_num, gamma, tau
        self.phi, self.lmbda, self.beta = phi, lmbda, beta
        self.qc_thres = cost_limit * (1 - gamma ** episode_len) / (1 - gamma) / episode_len  # bcql.py:99-100
        self.KP, self.KI, self.KD = PID_gains
        self.error_old, self.error_integral = 0.0, 0.0  # net.py:356-387
        idx = lambda grp, which: sorted({k.split(".")[2] for k in p if k.startswith(f"{grp}.{which}.")}, key=int)  # noqa: E731
        self.nets = {g: ([f"{g}.q1_nets.{i}" for i in idx(g, "q1_nets")], [f"{g}.q2_nets.{i}" for i in idx(g, "q2_nets")])
                     for g in ("critic", "cost_critic", "critic_old", "cost_critic_old")}
        self.opt_actor = torch.optim.Adam(_leafs(p, "actor"), lr=actor_lr)
        self.opt_critic = torch.optim.Adam(_leafs(p, "critic"), lr=critic_lr)
        self.opt_cost = torch.optim.Adam(_leafs(p, "cost_critic"), lr=critic_lr)
        self.opt_vae = torch.optim.Adam(_leafs(p, "vae"), lr=vae_lr)

    def _decode(self, obs, z):  # net.py:332-339 (z given: clamped to +-0.5 where it is drawn)
        p = self.p
        h = F.relu(F.linear(torch.cat([obs, z], 1), p["vae.d1.weight"], p["vae.d1.bias"]))
        h = F.relu(F.linear(h, p["vae.d2.weight"], p["vae.d2.bias"]))
        return self.max_action * torch.tanh(F.linear(h, p["vae.d3.weight"], p["vae.d3.bias"]))

    def _vae(self, obs, act, eps):  # net.py:319-330
        p = self.p
        h = F.relu(F.linear(torch.cat([obs, act], 1), p["vae.e1.weight"], p["vae.e1.bias"]))
        h = F.relu(F.linear(h, p["vae.e2.weight"], p["vae.e2.bias"]))
        mean = F.linear(h, p["vae.mean.weight"], p["vae.mean.bias"])
        std = torch.exp(torch.clamp(F.linear(h, p["vae.log_std.weight"], p["vae.log_std.bias"]), -4, 15))
        return self._decode(obs, mean + std * eps), mean, std

    def _perturb(self, prefix, obs, act):  # net.py:59-62
        a = self.phi * self.max_action * _mlp(self.p, prefix, torch.cat([obs, act], 1), torch.tanh, torch.tanh)
        return (a + act).clamp(-self.max_action, self.max_action)

    def _q_lists(self, grp, obs, act):  # EnsembleDoubleQCritic.predict, net.py:262-287
        x = torch.cat([obs, act], 1)
        l1 = [_mlp(self.p, s, x, F.relu).squeeze(-1) for s in self.nets[grp][0]]
        l2 = [_mlp(self.p, s, x, F.relu).squeeze(-1) for s in self.nets[grp][1]]
        return torch.stack(l1).min(0).values, torch.stack(l2).min(0).values, l1, l2

    def _target(self, grp_old, nobs, z):  # bcql.py:138-146
        B = nobs.shape[0]
        obs_n = torch.repeat_interleave(nobs, self.N, 0)
        a = self._perturb("actor_old.pi", obs_n, self._decode(obs_n, z.clamp(-0.5, 0.5)))
        q1, q2, _, _ = self._q_lists(grp_old, obs_n, a)
        q = self.lmbda * torch.min(q1, q2) + (1.0 - self.lmbda) * torch.max(q1, q2)
        return q.reshape(B, -1).max(1).values

    def _polyak(self, tgt, src):
        with torch.no_grad():
            for k, v in self.p.items():
                if k.startswith(tgt + "."):
                    v.mul_(1 - self.tau).add_(self.p[src + k[len(tgt):]], alpha=self.tau)

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done, noise) -> Dict[str, float]:
        obs, nobs, act, rew, cost, done = (_t(a) for a in (observations, next_observations, actions, rewards, costs, done))
        nz = {k: _t(v) for k, v in noise.items()}
        g, stats = self.gamma, {}
        # vae_loss  bcql.py:122-133
        recon, mean, std = self._vae(obs, act, nz["eps_vae"])
        loss_vae = F.mse_loss(recon, act) + self.beta * (-0.5 * (1 + torch.log(std.pow(2)) - mean.pow(2) - std.pow(2))).mean()
        self.opt_vae.zero_grad()
        loss_vae.backward()
        self.opt_vae.step()
        stats["loss/loss_vae"] = loss_vae.item()
        # critic_loss / cost_critic_loss  bcql.py:134-179
        for grp, opt, key, zkey in (("critic", self.opt_critic, "loss/critic_loss", "z_c"),
                                    ("cost_critic", self.opt_cost, "loss/cost_critic_loss", "z_cc")):
            with torch.no_grad():
                q_t = self._target(grp + "_old", nobs, nz[zkey])
                backup = rew + g * (1 - done) * q_t if grp == "critic" else cost + g * q_t
            _, _, l1, l2 = self._q_lists(grp, obs, act)
            loss = sum(F.mse_loss(q, backup) for q in l1 + l2)
            opt.zero_grad()
            loss.backward()
            opt.step()
            stats[key] = loss.item()
        # actor_loss  bcql.py:181-216 (critics / vae frozen: their .grad is discarded by their next zero_grad)
        a = self._perturb("actor.pi", obs, self._decode(obs, nz["z_actor"].clamp(-0.5, 0.5)))
        q1, q2, _, _ = self._q_lists("critic", obs, a)
        c1, c2, _, _ = self._q_lists("cost_critic", obs, a)
        q_pi, qc_pi = torch.min(q1, q2), torch.min(c1, c2)
        with torch.no_grad():  # LagrangianPIDController.control, net.py:370-387
            e = float((qc_pi - self.qc_thres).mean())
            d = max(e - self.error_old, 0.0)
            self.error_integral = max(self.error_integral + e, 0.0)
            self.error_old = e
            mult = max(self.KP * max(e, 0.0) + self.KI * self.error_integral + self.KD * d, 0.0)
        qc_penalty = ((qc_pi - self.qc_thres) * mult).mean()
        loss_a = -q_pi.mean() + qc_penalty
        self.opt_actor.zero_grad()
        loss_a.backward()
        self.opt_actor.step()
        stats.update({"loss/actor_loss": loss_a.item(), "loss/qc_penalty": qc_penalty.item(), "loss/lagrangian": mult})
        self._polyak("critic_old", "critic")
        self._polyak("cost_critic_old", "cost_critic")
        self._polyak("actor_old", "actor")
        return stats


# --------------------------------------------------------------------------------------------------------------------
class TorchCDT:
    def __init__(self, params, *, seq_len: int, num_heads: int, num_layers: int, cost_transform: bool = True,
                 stochastic: bool = True, init_temperature: float = 0.1, target_entropy: Optional[float] = None,
                 learning_rate: float = 1e-4, weight_decay: float = 1e-4, betas=(0.9, 0.999), clip_grad: Optional[float] = 0.25,
                 lr_warmup_steps: int = 500, loss_cost_weight: float = 0.02, loss_state_weight: float = 0.0,
                 dropout: float = 0.0):
        self.p = {k: torch.tensor(v, dtype=torch.float32) for k, v in params.items() if "causal_mask" not in k}
        for v in self.p.values():
            v.requires_grad_(True)
        self.T, self.H, self.NL = seq_len, num_heads, num_layers
        self.E = self.p["emb_norm.weight"].shape[0]
        self.cost_transform, self.stochastic, self.dropout = cost_transform, stochastic, dropout
        self.log_temperature = torch.tensor(math.log(init_temperature), requires_grad=True)  # cdt.py:144
        self.target_entropy = target_entropy
        self.lr, self.warmup, self.clip = learning_rate, lr_warmup_steps, clip_grad
        self.cw, self.sw = loss_cost_weight, loss_state_weight
        self.opt = torch.optim.AdamW(list(self.p.values()), lr=learning_rate, weight_decay=weight_decay, betas=betas)
        self.opt_T = torch.optim.Adam([self.log_temperature], lr=1e-4, betas=(0.9, 0.999))  # cdt.py:332-337
        self.steps = 0

    def _drop(self, x):
        return F.dropout(x, self.dropout, True) if self.dropout > 0 else x

    def forward(self, states, actions, returns, costs_to_go, time_steps, mask):
        p, E, H = self.p, self.E, self.H
        B, T, _ = states.shape
        R, S, d = 4, 4 * T, E // H
        te = p["timestep_emb.weight"][time_steps]  # cdt.py:180-183
        ctg = (50.0 - costs_to_go) if self.cost_transform else costs_to_go  # cdt.py:78-81,187-188
        s_e = F.linear(states, p["state_emb.weight"], p["state_emb.bias"]) + te
        a_e = F.linear(actions, p["action_emb.weight"], p["action_emb.bias"]) + te
        c_e = F.linear(ctg[..., None], p["cost_emb.weight"], p["cost_emb.bias"]) + te
        r_e = F.linear(returns[..., None], p["return_emb.weight"], p["return_emb.bias"]) + te
        seq = torch.stack([r_e, c_e, s_e, a_e], 2).reshape(B, S, E)  # (r, c, s, a) per timestep  cdt.py:185-200
        key_pad = torch.repeat_interleave(mask <= 0, R, dim=1)  # True = ignore  cdt.py:202-205
        x = self._drop(F.layer_norm(seq, (E,), p["emb_norm.weight"], p["emb_norm.bias"]))  # cdt.py:220-222
        blocked = torch.triu(torch.ones(S, S, dtype=torch.bool), 1)[None, None] | key_pad[:, None, None, :]
        for l in range(self.NL):  # TransformerBlock.forward  net.py:422-441
            pre = f"blocks.{l}."
            n1 = F.layer_norm(x, (E,), p[pre + "norm1.weight"], p[pre + "norm1.bias"])
            qkv = F.linear(n1, p[pre + "attention.in_proj_weight"], p[pre + "attention.in_proj_bias"])
            q, k, v = (t.reshape(B, S, H, d).transpose(1, 2) for t in qkv.split(E, -1))
            sc = (q @ k.transpose(-1, -2)) / math.sqrt(d)
            P = self._drop(torch.softmax(sc.masked_fill(blocked, float("-inf")), -1))  # attention-probability dropout
            o = (P @ v).transpose(1, 2).reshape(B, S, E)
            x = x + self._drop(F.linear(o, p[pre + "attention.out_proj.weight"], p[pre + "attention.out_proj.bias"]))
            n2 = F.layer_norm(x, (E,), p[pre + "norm2.weight"], p[pre + "norm2.bias"])
            h = F.gelu(F.linear(n2, p[pre + "mlp.0.weight"], p[pre + "mlp.0.bias"]))
            x = x + self._drop(F.linear(h, p[pre + "mlp.2.weight"], p[pre + "mlp.2.bias"]))
        out = F.layer_norm(x, (E,), p["out_norm.weight"], p["out_norm.bias"]).reshape(B, T, R, E)
        sf, af = out[:, :, 2], out[:, :, 3]  # the action head reads the STATE token  cdt.py:239-240
        res = {}
        if self.stochastic:
            res["mu"] = F.linear(sf, p["action_head.mu.weight"], p["action_head.mu.bias"])
            res["ls"] = F.linear(sf, p["action_head.log_std.weight"], p["action_head.log_std.bias"])
        else:
            res["act"] = F.linear(sf, p["action_head.0.weight"], p["action_head.0.bias"])
        res["cost_logp"] = F.log_softmax(F.linear(af, p["cost_pred_head.weight"], p["cost_pred_head.bias"]), -1)
        res["state_pred"] = F.linear(af, p["state_pred_head.weight"], p["state_pred_head.bias"])
        return res

    def train_one_step(self, states, actions, returns, costs_return, time_steps, mask, episode_cost, costs) -> Dict[str, float]:
        states, actions, returns, costs_return, mask = (_t(a) for a in (states, actions, returns, costs_return, mask))
        time_steps = torch.as_tensor(time_steps, dtype=torch.int64)
        costs_i = torch.as_tensor(costs).to(torch.int64)
        res = self.forward(states, actions, returns, costs_return, time_steps, mask)
        valid = mask > 0
        stats = {}
        if self.stochastic:  # cdt.py:357-372
            mu, ls = res["mu"], res["ls"]
            dist = torch.distributions.Normal(mu, ls.exp())
            ll = dist.log_prob(actions)[valid].mean()
            ent = dist.entropy()[valid].mean()
            temp = self.log_temperature.exp().detach()
            act_loss = -(ll + temp * ent)
            stats.update(nll=-ll.item(), ent=ent.item(), ent_reg=temp.item())
        else:
            act_loss = (F.mse_loss(res["act"], actions, reduction="none") * mask[..., None]).mean()
        cost_loss = (F.nll_loss(res["cost_logp"].reshape(-1, 2), costs_i.reshape(-1), reduction="none")
                     * mask.reshape(-1)).mean()  # mean over ALL B*T  cdt.py:378-380
        acc = ((res["cost_logp"].argmax(-1) == costs_i) * mask).sum() / mask.sum()
        state_loss = (F.mse_loss(res["state_pred"][:, :-1], states[:, 1:], reduction="none") * mask[:, :-1, None]).mean()
        loss = act_loss + self.cw * cost_loss + self.sw * state_loss
        self.opt.zero_grad()
        loss.backward()
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(list(self.p.values()), self.clip)
        for grp in self.opt.param_groups:  # LambdaLR warm-up  cdt.py:327-330
            grp["lr"] = self.lr * min((self.steps + 1) / self.warmup, 1.0)
        self.opt.step()
        if self.stochastic:  # cdt.py:402-407
            self.opt_T.zero_grad()
            (self.log_temperature.exp() * (ent.detach() - self.target_entropy)).backward()
            self.opt_T.step()
        self.steps += 1
        stats.update(all_loss=loss.item(), act_loss=act_loss.item(), cost_loss=cost_loss.item(), cost_acc=acc.item(),
                     state_loss=state_loss.item(), train_lr=self.lr * min((self.steps + 1) / self.warmup, 1.0))
        return stats
