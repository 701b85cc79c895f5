"""Constrained Decision Transformer on MI355X behind the reference's API (osrl/algorithms/cdt.py).

``CDT`` keeps the constructor and ``state_dict`` layout of cdt.py:45-164 (the same torch container modules
are created in the same order, so a seeded construction reproduces the reference's initial weights);
``CDTTrainer.train_one_step`` keeps the signature of cdt.py:343.  Supported configuration = the reference's
every constructor variant of cdt.py:45-141: any subset of the return / cost tokens, with or without the timestep
embedding, the cost-prefix token, the add / mul / cat cost features on the state feature, deeper action heads,
stochastic or deterministic, dropout.  Limits: <= 128 tokens per sequence, embedding_dim <= 256.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..common.logger import store_stats
from ..common.net import DiagGaussianActor, TransformerBlock, bind_group, mlp, plan_group
from ..engine.core import FlatGroup, require_cuda


class _Params(nn.Module):
    """Holds the reference-named submodules so that plan_group/bind_group see keys without a prefix."""


class CDT(nn.Module):
    def __init__(self, state_dim: int, action_dim: int, max_action: float, seq_len: int = 10,
                 episode_len: int = 1000, embedding_dim: int = 128, num_layers: int = 4, num_heads: int = 8,
                 attention_dropout: float = 0.0, residual_dropout: float = 0.0, embedding_dropout: float = 0.0,
                 time_emb: bool = True, use_rew: bool = False, use_cost: bool = False, cost_transform: bool = False,
                 add_cost_feat: bool = False, mul_cost_feat: bool = False, cat_cost_feat: bool = False,
                 action_head_layers: int = 1, cost_prefix: bool = False, stochastic: bool = False,
                 init_temperature=0.1, target_entropy=None, device: str = "cuda"):
        super().__init__()
        unsupported = []
        if not all(0.0 <= p < 1.0 for p in (attention_dropout, residual_dropout, embedding_dropout)):
            raise ValueError("dropout probabilities must be in [0, 1)")
        if action_head_layers < 1:
            raise ValueError("action_head_layers must be >= 1")
        seq_repeat = 2 + int(bool(use_cost)) + int(bool(use_rew))  # cdt.py:96-105
        S = seq_repeat * seq_len + int(bool(cost_prefix))           # cdt.py:107-112
        if embedding_dim % num_heads or embedding_dim > 512 or 4 * embedding_dim > 1024 or S > 128:
            unsupported.append("embedding_dim > 256 or more than 128 tokens per sequence")
        else:  # the attention backward keeps two [S, head_dim] tiles, row statistics and the keep flags in LDS
            r16 = lambda x: (x + 15) // 16 * 16  # noqa: E731
            sp, dp = r16(S), r16(embedding_dim // num_heads)
            if 4 * (2 * sp * (dp + 8) + 4 * sp) + sp * sp > 160 * 1024:
                unsupported.append(f"{S} tokens with head_dim {embedding_dim // num_heads} "
                                   "(attention backward tiles > 160 KB LDS)")
        if unsupported:
            raise NotImplementedError("osrl_amd CDT does not support: " + "; ".join(unsupported))
        self.seq_len, self.embedding_dim = seq_len, embedding_dim
        self.state_dim, self.action_dim = state_dim, action_dim
        self.episode_len, self.max_action = episode_len, max_action
        self.num_layers, self.num_heads = num_layers, num_heads
        self.cost_transform_on = bool(cost_transform)
        self.cost_transform = (lambda x: 50 - x) if cost_transform else None
        # cdt.py:243-250: every cost-feature op also requires use_cost
        self.add_cost_feat = bool(add_cost_feat and use_cost)
        self.mul_cost_feat = bool(mul_cost_feat and use_cost)
        self.cat_cost_feat = bool(cat_cost_feat and use_cost)
        self.stochastic = stochastic
        self.attention_dropout, self.residual_dropout = float(attention_dropout), float(residual_dropout)
        self.embedding_dropout = float(embedding_dropout)
        self.time_emb, self.use_rew, self.use_cost = bool(time_emb), bool(use_rew), bool(use_cost)
        self.cost_prefix = bool(cost_prefix)
        self.seq_repeat = seq_repeat
        self.action_head_layers = int(action_head_layers)
        self.device = str(device)
        dev = require_cuda(device)

        # container modules, created in the reference's order (cdt.py:87-141)
        self.emb_drop = nn.Dropout(embedding_dropout)
        self.emb_norm = nn.LayerNorm(embedding_dim)
        self.out_norm = nn.LayerNorm(embedding_dim)
        if self.time_emb:
            self.timestep_emb = nn.Embedding(episode_len + seq_len, embedding_dim)
        self.state_emb = nn.Linear(state_dim, embedding_dim)
        self.action_emb = nn.Linear(action_dim, embedding_dim)
        if self.use_cost:
            self.cost_emb = nn.Linear(1, embedding_dim)
        if self.use_rew:
            self.return_emb = nn.Linear(1, embedding_dim)
        if self.cost_prefix:
            self.prefix_emb = nn.Linear(1, embedding_dim)
        self.blocks = nn.ModuleList([TransformerBlock(S, embedding_dim, num_heads, attention_dropout,
                                                      residual_dropout) for _ in range(num_layers)])
        # cdt.py:125 tests the RAW constructor flag: the head is 2E wide whenever cat_cost_feat was asked for
        Eh = self.head_in_dim = 2 * embedding_dim if cat_cost_feat else embedding_dim
        if cat_cost_feat and not use_cost:
            raise NotImplementedError("cat_cost_feat without use_cost builds a 2E-wide head the reference's forward "
                                      "feeds E-wide features (cdt.py:125,247-250): it cannot run there either")
        if stochastic:  # cdt.py:127-133
            if action_head_layers >= 2:
                self.action_head = nn.Sequential(nn.Linear(Eh, Eh), nn.GELU(), DiagGaussianActor(Eh, action_dim))
                self.head_hidden_keys, self.head_out_key = ["cdt.action_head.0.weight"], "cdt.action_head.2.head.weight"
            else:
                self.action_head = DiagGaussianActor(Eh, action_dim)
                self.head_hidden_keys, self.head_out_key = [], "cdt.action_head.head.weight"
        else:  # cdt.py:134-137
            self.action_head = mlp([Eh] * action_head_layers + [action_dim], activation=nn.GELU,
                                   output_activation=nn.Identity)
            self.head_hidden_keys = [f"cdt.action_head.{2 * i}.weight" for i in range(action_head_layers - 1)]
            self.head_out_key = f"cdt.action_head.{2 * (action_head_layers - 1)}.weight"
        self.state_pred_head = nn.Linear(embedding_dim, state_dim)
        self.cost_pred_head = nn.Linear(embedding_dim, 2)
        self.apply(self._init_weights)

        g = FlatGroup("cdt", dev)
        plan_group(g, "cdt", self)
        g.finalize()
        bind_group(g, "cdt", self)
        self.groups: Dict[str, FlatGroup] = {"cdt": g}
        if stochastic:  # cdt.py:143-146: a plain tensor outside the state_dict
            self.log_temperature = torch.full((1,), float(np.log(init_temperature)), dtype=torch.float32, device=dev)
            self.target_entropy = target_entropy
        self._engine = None

    @staticmethod
    def _init_weights(module: nn.Module):
        """cdt.py:156-164."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if isinstance(module, nn.Linear) and module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            torch.nn.init.zeros_(module.bias)
            torch.nn.init.ones_(module.weight)

    def temperature(self):
        return self.log_temperature.exp() if self.stochastic else None

    def repack(self) -> None:
        for g in self.groups.values():
            if g.device.type == "cuda":
                g.repack()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        if assign:
            raise RuntimeError("assign=True would detach parameters from their flat HBM group")
        res = super().load_state_dict(state_dict, strict=strict)
        self.repack()
        return res

    def _apply(self, fn, *a, **k):
        raise RuntimeError("osrl_amd models are bound to their HIP device at construction (pass device=)")

    def engine(self, batch_size: int, cfg: Optional[dict] = None, dist=None):
        from ..common.checkpoint import engine_handoff
        from ..engine.cdt import CDTEngine
        if self._engine is None or self._engine.B != batch_size or dist is not None or \
                (cfg is not None and cfg != self._engine.cfg):
            if cfg is None:
                raise RuntimeError("build a CDTTrainer before training")
            old, self._engine = self._engine, CDTEngine(self, batch_size, cfg, dist=dist)
            engine_handoff(self, self._engine, old)
        return self._engine

    @torch.no_grad()
    def forward(self, states, actions, returns_to_go, costs_to_go, time_steps, padding_mask=None,
                episode_cost=None):
        """Inference forward (cdt.py:166-265): returns (action_preds, cost_preds, state_preds) where
        action_preds is ``torch.distributions.Normal(mu, std)`` for a stochastic head."""
        B = states.shape[0]
        cfg = self._engine.cfg if self._engine is not None else dict(
            learning_rate=1e-4, weight_decay=1e-4, betas=(0.9, 0.999), clip_grad=0.25, lr_warmup_steps=1,
            loss_cost_weight=0.0, loss_state_weight=0.0, no_entropy=False)
        from ..engine.cdt import CDTEngine
        Tin, T = states.shape[1], self.seq_len
        if Tin > T:
            raise ValueError(f"window of {Tin} steps > seq_len {T}")
        if getattr(self, "_infer", None) is None or self._infer.B != B:
            self._infer = CDTEngine(self, B, cfg, inference=True)
        e = self._infer
        mask = torch.ones(B, Tin, device=states.device) if padding_mask is None else \
            (~padding_mask.to(torch.bool)).float()
        if Tin < T:
            # shorter windows (early rollout steps, cdt.py:485-489): left-align and zero-pad the tail; with the
            # causal mask the first Tin positions are exactly the short-sequence result
            def pad(x, val=0):
                out = torch.full((B, T) + tuple(x.shape[2:]), val, dtype=x.dtype, device=x.device)
                out[:, :Tin] = x
                return out
            states, actions, returns_to_go, costs_to_go = pad(states), pad(actions), pad(returns_to_go), pad(costs_to_go)
            time_steps, mask = pad(time_steps), pad(mask)
        if self.cost_prefix and episode_cost is None:
            raise ValueError("cost_prefix=True: pass episode_cost [B] (cdt.py:207-213)")
        e.load_batch(states, actions, returns_to_go, costs_to_go, time_steps, mask, torch.zeros_like(mask),
                     torch.as_tensor(episode_cost, dtype=torch.float32, device=states.device).reshape(B)
                     if self.cost_prefix else None)
        self.repack()
        if self.training and max(self.attention_dropout, self.residual_dropout, self.embedding_dropout) > 0:
            e.st.tick()  # a model left in train() mode draws a fresh dropout mask per call, like nn.Dropout
        e.forward(train=self.training)
        ad, od = self.action_dim, self.state_dim
        if self.stochastic:
            mu = e.head[:, :ad].reshape(B, T, ad)[:, :Tin].clone()
            ls = e.head[:, ad:].reshape(B, T, ad)[:, :Tin].clone()
            ap = torch.distributions.Normal(mu, ls.exp())
        else:
            ap = e.head.reshape(B, T, ad)[:, :Tin].clone()
        return (ap, torch.log_softmax(e.logits.reshape(B, T, 2)[:, :Tin], -1),
                e.sp.reshape(B, T, od)[:, :Tin].clone())


class CDTTrainer:
    """cdt.py:268-418."""

    def __init__(self, model: CDT, env=None, logger=None, learning_rate: float = 1e-4,
                 weight_decay: float = 1e-4, betas: Tuple[float, ...] = (0.9, 0.999), clip_grad: float = 0.25,
                 lr_warmup_steps: int = 10000, reward_scale: float = 1.0, cost_scale: float = 1.0,
                 loss_cost_weight: float = 0.0, loss_state_weight: float = 0.0, cost_reverse: bool = False,
                 no_entropy: bool = False, device="cuda", stats_mode: str = "lazy", use_graph: bool = True,
                 seed: int = 0) -> None:
        self.model, self.logger, self.env = model, logger, env
        self.clip_grad, self.reward_scale, self.cost_scale, self.device = clip_grad, reward_scale, cost_scale, device
        self.cost_weight, self.state_weight = loss_cost_weight, loss_state_weight
        self.cost_reverse, self.no_entropy = cost_reverse, no_entropy
        self.stochastic = model.stochastic
        self.max_action = model.max_action
        self.stats_mode, self.use_graph = stats_mode, use_graph
        self.cfg = dict(learning_rate=learning_rate, weight_decay=weight_decay, betas=tuple(betas),
                        clip_grad=clip_grad, lr_warmup_steps=lr_warmup_steps, loss_cost_weight=loss_cost_weight,
                        loss_state_weight=loss_state_weight, no_entropy=no_entropy, seed=int(seed))

    def train_one_step(self, states, actions, returns, costs_return, time_steps, mask, episode_cost, costs):
        """cdt.py:343-418 (``episode_cost`` only feeds the cost-prefix variant)."""
        eng = self.model.engine(states.shape[0], self.cfg)
        eng.step(states, actions, returns, costs_return, time_steps, mask, costs, use_graph=self.use_graph,
                 episode_cost=episode_cost if self.model.cost_prefix else None)
        keys = None if self.stochastic else ["all_loss", "act_loss", "cost_loss", "cost_acc", "state_loss", "train_lr"]
        store_stats(self.logger, eng.st, self.stats_mode, tab="train", keys=keys)

    def evaluate(self, num_rollouts, target_return, target_cost):
        """cdt.py:420-434.  With a ``VecSyntheticSafeEnv`` as ``self.env`` the ``num_rollouts`` episodes run as one
        batch on device (engine/rollout.py ``CDTBatchedRollout``)."""
        from ..common.synthetic_env import VecSyntheticSafeEnv
        if isinstance(self.env, VecSyntheticSafeEnv):
            from ..engine.rollout import CDTBatchedRollout
            if self.env.E != num_rollouts:
                raise ValueError(f"the vector environment holds {self.env.E} episodes, evaluate() was asked for "
                                 f"{num_rollouts}")
            ro = getattr(self, "_rollout", None)
            if ro is None or ro[0] != id(self.env):
                ro = self._rollout = (id(self.env), CDTBatchedRollout(self.model, self.env, self.cost_scale,
                                                                      self.cost_reverse, self.use_graph))
            r, c, n = ro[1].run(target_return, target_cost)
            return float(r.mean()) / self.reward_scale, float(c.mean()) / self.cost_scale, float(n.mean())
        self.model.eval()
        rets, costs, lens = [], [], []
        for _ in range(num_rollouts):
            r, l, c = self.rollout(self.model, self.env, target_return, target_cost)
            rets.append(r)
            lens.append(l)
            costs.append(c)
        self.model.train()
        return np.mean(rets) / self.reward_scale, np.mean(costs) / self.cost_scale, np.mean(lens)

    @torch.no_grad()
    def rollout(self, model: CDT, env, target_return: float, target_cost: float):
        """cdt.py:436-518: autoregressive rollout on a sliding window of the last seq_len steps."""
        dev = torch.device(model.device)
        EL, T = model.episode_len, model.seq_len
        states = torch.zeros(1, EL + 1, model.state_dim, device=dev)
        actions = torch.zeros(1, EL, model.action_dim, device=dev)
        returns = torch.zeros(1, EL + 1, device=dev)
        costs = torch.zeros(1, EL + 1, device=dev)
        time_steps = torch.arange(EL, dtype=torch.long, device=dev).view(1, -1)
        obs, info = env.reset()
        states[:, 0] = torch.as_tensor(obs, device=dev)
        returns[:, 0] = float(target_return)
        costs[:, 0] = float(target_cost)
        epi_cost = torch.tensor([target_cost], dtype=torch.float, device=dev)
        ep_ret, ep_cost, ep_len = 0.0, 0.0, 0
        for step in range(EL):
            lo = max(0, step + 1 - T)
            acts, _, _ = model(states[:, lo:step + 1], actions[:, lo:step + 1], returns[:, lo:step + 1],
                               costs[:, lo:step + 1], time_steps[:, lo:step + 1], None, epi_cost)
            if self.stochastic:
                acts = acts.mean
            act = acts.clamp(-self.max_action, self.max_action)[0, -1].cpu().numpy()
            obs_next, reward, terminated, truncated, info = env.step(act)
            cost = ((1.0 - info["cost"]) if self.cost_reverse else info["cost"]) * self.cost_scale
            actions[:, step] = torch.as_tensor(act, device=dev)
            states[:, step + 1] = torch.as_tensor(obs_next, device=dev)
            returns[:, step + 1] = returns[:, step] - float(reward)
            costs[:, step + 1] = costs[:, step] - float(cost)
            ep_ret += reward
            ep_len += 1
            ep_cost += info["cost"]
            if terminated or truncated:
                break
        return ep_ret, ep_len, ep_cost
