"""CPU oracle for the Constrained Decision Transformer train step (TEST INFRASTRUCTURE ONLY --
see oracle/osrl_oracle.py for the rules; pinned by tests/golden/cdt_*.npz captured from the reference).

Restates, in numpy with a hand-derived backward pass:
  * CDT.forward                      osrl/algorithms/cdt.py:166-265
  * TransformerBlock.forward         osrl/common/net.py:422-441  (nn.MultiheadAttention recipe)
  * DiagGaussianActor.forward        osrl/common/net.py:530-533
  * CDTTrainer.train_one_step        osrl/algorithms/cdt.py:343-418  (AdamW + clip_grad_norm_ + LambdaLR warm-up
                                     + temperature Adam)
Supported configuration = the reference's train defaults (examples/configs/cdt_configs.py:22-89):
time_emb, use_rew, use_cost, optional cost_transform, action_head_layers=1, no cost-feature variants, no
cost prefix, stochastic or deterministic head.  Dropout: the caller passes the keep-multipliers (0 or 1/(1-p)) of
every nn.Dropout site explicitly (``drop``: 'emb' [B,S,E]; per layer 'attn{l}' [B,H,S,S], 'res1_{l}', 'res2_{l}'
[B,S,E]); ``None`` = eval mode / p = 0.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
from scipy.special import erf

from .osrl_oracle import Adam

Array = np.ndarray
State = Dict[str, Array]


def layer_norm(x: Array, g: Array, b: Array, eps: float = 1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)  # biased variance (nn.LayerNorm)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * g + b, (xhat, rstd)


def layer_norm_bwd(dy: Array, cache, g: Array):
    xhat, rstd = cache
    dxhat = dy * g
    dx = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xhat).sum(red), dy.sum(red)


def gelu(x: Array) -> Array:  # nn.GELU() exact erf form
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def gelu_grad(x: Array) -> Array:
    return 0.5 * (1.0 + erf(x / math.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


class OracleCDT:
    def __init__(self, params: State, *, seq_len: int, num_heads: int, num_layers: int, max_action: float = 1.0,
                 cost_transform: bool = True, stochastic: bool = True, init_temperature: float = 0.1,
                 target_entropy: Optional[float] = None, learning_rate: float = 1e-4, weight_decay: float = 1e-4,
                 betas=(0.9, 0.999), clip_grad: Optional[float] = 0.25, lr_warmup_steps: int = 500,
                 loss_cost_weight: float = 0.02, loss_state_weight: float = 0.0, no_entropy: bool = False,
                 time_emb: bool = True, use_rew: bool = True, use_cost: bool = True, add_cost_feat: bool = False,
                 mul_cost_feat: bool = False, cat_cost_feat: bool = False, action_head_layers: int = 1,
                 cost_prefix: bool = False, dtype=np.float32):
        self.p = {k: np.array(v, dtype=dtype) for k, v in params.items() if "causal_mask" not in k}
        self.dtype = dtype
        self.T, self.H, self.NL = seq_len, num_heads, num_layers
        self.E = self.p["emb_norm.weight"].shape[0]
        self.cost_transform, self.stochastic = cost_transform, stochastic
        self.max_action = max_action
        self.log_temperature = math.log(init_temperature)  # cdt.py:144
        self.target_entropy = target_entropy
        self.lr, self.warmup, self.clip = learning_rate, lr_warmup_steps, clip_grad
        self.cw, self.sw, self.no_entropy = loss_cost_weight, loss_state_weight, no_entropy
        self.opt = Adam(list(self.p.keys()), learning_rate, betas[0], betas[1], 1e-8, weight_decay)  # AdamW :321-326
        self.opt_T = Adam(["log_temperature"], 1e-4, 0.9, 0.999)  # cdt.py:332-337
        self.steps = 0
        # constructor variants (cdt.py:58-66,96-141): token set, cost features on the state feature, head depth, prefix
        self.time_emb, self.use_rew, self.use_cost, self.prefix = time_emb, use_rew, use_cost, cost_prefix
        self.R = 2 + int(use_rew) + int(use_cost)
        self.add_cf, self.mul_cf, self.cat_cf = (add_cost_feat and use_cost, mul_cost_feat and use_cost,
                                                 cat_cost_feat and use_cost)  # cdt.py:243-250 all require use_cost
        self.head_layers = action_head_layers

    # ---- action head (cdt.py:127-137): hidden Linear+GELU layers, then the output layer(s)
    def _head_keys(self):
        if self.stochastic:
            hidden = ["action_head.0"] if self.head_layers >= 2 else []
            pre = "action_head.2." if self.head_layers >= 2 else "action_head."
            return hidden, [pre + "mu", pre + "log_std"]
        n = self.head_layers
        return [f"action_head.{2 * i}" for i in range(n - 1)], [f"action_head.{2 * (n - 1)}"]

    # ------------------------------------------------------------------ forward
    def forward(self, states, actions, returns, costs_to_go, time_steps, mask, drop=None, episode_cost=None):
        drop = drop or {}
        one = self.dtype(1.0)
        p, E, H, R = self.p, self.E, self.H, self.R
        B, T, _ = states.shape
        S = R * T + int(self.prefix)
        c = {}
        te = p["timestep_emb.weight"][time_steps] if self.time_emb else self.dtype(0.0)  # [B,T,E]  cdt.py:180-183
        ctg = (50.0 - costs_to_go) if self.cost_transform else costs_to_go  # cdt.py:78-81,187-188
        c["ctg"] = ctg
        s_e = states @ p["state_emb.weight"].T + p["state_emb.bias"] + te
        a_e = actions @ p["action_emb.weight"].T + p["action_emb.bias"] + te
        toks = [s_e, a_e]
        c_e = None
        if self.use_cost:  # cdt.py:190-192: costs go in front of the state token ...
            c_e = ctg[..., None] * p["cost_emb.weight"][:, 0] + p["cost_emb.bias"] + te
            toks.insert(0, c_e)
        if self.use_rew:  # cdt.py:193-195: ... and returns in front of those
            toks.insert(0, returns[..., None] * p["return_emb.weight"][:, 0] + p["return_emb.bias"] + te)
        c["c_e"] = c_e
        seq = np.stack(toks, 2).reshape(B, R * T, E)  # (r,c,s,a) per timestep  cdt.py:185-200
        key_pad = np.repeat(mask <= 0, R, axis=1)  # [B,S] True = ignore  cdt.py:202-205
        if self.prefix:  # cdt.py:207-218: one token in front, from the episode's cost budget; masked like token 0
            pe = np.asarray(episode_cost, self.dtype)[:, None, None] * p["prefix_emb.weight"][:, 0] + p["prefix_emb.bias"]
            seq = np.concatenate([pe, seq], 1)
            key_pad = np.concatenate([key_pad[:, :1], key_pad], 1)
        x, c["ln_emb"] = layer_norm(seq, p["emb_norm.weight"], p["emb_norm.bias"])
        x = x * drop.get("emb", one)  # emb_drop  cdt.py:222
        causal = np.triu(np.ones((S, S), bool), 1)  # True = blocked  net.py:417-418
        blocked = causal[None, None] | key_pad[:, None, None, :]
        d = E // H
        c["blocks"] = []
        for l in range(self.NL):
            pre = f"blocks.{l}."
            bc = {"x_in": x}
            n1, bc["ln1"] = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
            qkv = n1 @ p[pre + "attention.in_proj_weight"].T + p[pre + "attention.in_proj_bias"]
            q, k, v = (qkv[..., i * E:(i + 1) * E].reshape(B, S, H, d).transpose(0, 2, 1, 3) for i in range(3))
            sc = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(d)
            sc = np.where(blocked, -np.inf, sc)
            sc = sc - sc.max(-1, keepdims=True)
            P = np.exp(sc)
            P = P / P.sum(-1, keepdims=True)
            Pd = P * drop.get(f"attn{l}", one)  # attention-probability dropout inside nn.MultiheadAttention
            o = (Pd @ v).transpose(0, 2, 1, 3).reshape(B, S, E)
            att = o @ p[pre + "attention.out_proj.weight"].T + p[pre + "attention.out_proj.bias"]
            x = x + att * drop.get(f"res1_{l}", one)  # net.py:439
            n2, bc["ln2"] = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
            hpre = n2 @ p[pre + "mlp.0.weight"].T + p[pre + "mlp.0.bias"]
            h = gelu(hpre)
            x = x + (h @ p[pre + "mlp.2.weight"].T + p[pre + "mlp.2.bias"]) * drop.get(f"res2_{l}", one)  # net.py:414
            bc.update(n1=n1, q=q, k=k, v=v, P=P, Pd=Pd, o=o, n2=n2, hpre=hpre, h=h)
            c["blocks"].append(bc)
        out, c["ln_out"] = layer_norm(x, p["out_norm.weight"], p["out_norm.bias"])
        if self.prefix:
            out = out[:, 1:]  # cdt.py:229-231
        out4 = out.reshape(B, T, R, E)
        sf, af = out4[:, :, R - 2], out4[:, :, R - 1]  # action head reads the STATE token  cdt.py:239-240
        feat = sf  # cdt.py:243-250: the (detached) cost embedding joins the state feature
        if self.add_cf:
            feat = feat + c_e
        if self.mul_cf:
            feat = feat * c_e
        if self.cat_cf:
            feat = np.concatenate([feat, c_e], -1)
        hidden, outs = self._head_keys()
        hcache, h = [], feat
        for k in hidden:
            pre_h = h @ p[k + ".weight"].T + p[k + ".bias"]
            hcache.append((h, pre_h))
            h = gelu(pre_h)
        c.update(sf=sf, af=af, head_in=h, head_hidden=hcache)
        res = {}
        if self.stochastic:
            res["mu"] = h @ p[outs[0] + ".weight"].T + p[outs[0] + ".bias"]
            res["ls"] = h @ p[outs[1] + ".weight"].T + p[outs[1] + ".bias"]
        else:
            res["act"] = h @ p[outs[0] + ".weight"].T + p[outs[0] + ".bias"]
        logits = af @ p["cost_pred_head.weight"].T + p["cost_pred_head.bias"]
        z = logits - logits.max(-1, keepdims=True)
        res["cost_logp"] = z - np.log(np.exp(z).sum(-1, keepdims=True))
        res["state_pred"] = af @ p["state_pred_head.weight"].T + p["state_pred_head.bias"]
        return res, c

    def act_mean(self, states, actions, returns, costs_to_go, time_steps, mask, episode_cost=None):
        """Deterministic action prediction (mean) for every timestep of the window."""
        res, _ = self.forward(*(np.asarray(a, self.dtype) if a.dtype.kind == "f" else a
                                for a in (states, actions, returns, costs_to_go, time_steps, mask)),
                              episode_cost=episode_cost)
        return res["mu"] if self.stochastic else res["act"]

    # ------------------------------------------------------------------ one train step
    def train_one_step(self, states, actions, returns, costs_return, time_steps, mask, episode_cost, costs,
                       drop=None, norm=None, grads_only=False):
        """``norm`` (test aid for batches too big for one host pass): the batch-GLOBAL normalisers
        ``dict(nv=valid tokens * action_dim, bt=B*T, state_n=B*(T-1)*state_dim, act_n=B*T*action_dim)`` to use instead
        of this call's own -- the loss is a sum over samples once they are fixed, so the gradient of a big batch is the
        sum of ``grads_only=True`` calls over its chunks (returns the unclipped gradient dict, no update)."""
        dt = self.dtype
        drop = {k: np.asarray(v, dt) for k, v in (drop or {}).items()}
        one = dt(1.0)
        states, actions, returns, costs_return, mask = (np.asarray(a, dt) for a in
                                                        (states, actions, returns, costs_return, mask))
        time_steps = np.asarray(time_steps, np.int64)
        costs_i = np.asarray(costs).astype(np.int64)
        p, E, H = self.p, self.E, self.H
        B, T, od = states.shape
        ad = actions.shape[-1]
        R = self.R
        S, d = R * T + int(self.prefix), E // H
        res, c = self.forward(states, actions, returns, costs_return, time_steps, mask, drop, episode_cost)
        valid = mask > 0
        nv = max(int(valid.sum()), 1) * ad
        n_bt, n_state, n_act = B * T, None, None
        if norm is not None:
            nv, n_bt, n_state, n_act = norm["nv"], norm["bt"], norm["state_n"], norm["act_n"]
        stats = {}
        g: State = {}
        # ---- losses (cdt.py:357-394) and gradients wrt the head outputs
        temp = math.exp(self.log_temperature)
        if self.stochastic:
            mu, ls = res["mu"], res["ls"]
            std = np.exp(ls)
            zz = (actions - mu) / std
            logp = -0.5 * zz * zz - ls - 0.5 * math.log(2 * math.pi)
            ll = (logp * valid[..., None]).sum() / nv
            ent = ((0.5 + 0.5 * math.log(2 * math.pi) + ls) * valid[..., None]).sum() / nv
            ent_reg = 0.0 if self.no_entropy else temp
            act_loss = -(ll + ent_reg * ent)
            dmu = -(zz / std) * valid[..., None] / nv            # d(-ll)/dmu
            dls = (-(zz * zz - 1.0) - ent_reg) * valid[..., None] / nv  # d(-ll - reg*ent)/dls
            stats.update(nll=-ll, ent=ent, ent_reg=ent_reg)
        else:
            pred = res["act"]
            act_loss = (((pred - actions) ** 2) * mask[..., None]).mean()
            dact = 2 * (pred - actions) * mask[..., None] / (pred.size if n_act is None else n_act)
        lp = res["cost_logp"]
        onehot = np.eye(2, dtype=dt)[costs_i]
        cost_loss = (-(lp * onehot).sum(-1) * mask).mean()  # mean over ALL B*T  cdt.py:378-380
        dlogits = (np.exp(lp) - onehot) * mask[..., None] / n_bt * self.cw
        pred_c = lp.argmax(-1)
        acc = ((pred_c == costs_i) * mask).sum() / mask.sum()
        sp = res["state_pred"]
        diff = sp[:, :-1] - states[:, 1:]
        state_loss = ((diff ** 2) * mask[:, :-1, None]).mean()
        dsp = np.zeros_like(sp)
        dsp[:, :-1] = 2 * diff * mask[:, :-1, None] / (diff.size if n_state is None else n_state) * self.sw
        loss = act_loss + self.cw * cost_loss + self.sw * state_loss

        # ---- heads backward
        sf, af = c["sf"], c["af"]
        f2 = lambda a: a.reshape(-1, a.shape[-1])  # noqa: E731
        hidden, outs = self._head_keys()
        hin = c["head_in"]
        dh = np.zeros_like(hin)
        for k, dd in zip(outs, (dmu, dls) if self.stochastic else (dact,)):
            g[k + ".weight"] = f2(dd).T @ f2(hin)
            g[k + ".bias"] = f2(dd).sum(0)
            dh = dh + dd @ p[k + ".weight"]
        for k, (h_in, pre_h) in zip(reversed(hidden), reversed(c["head_hidden"])):
            dpre = dh * gelu_grad(pre_h)
            g[k + ".weight"] = f2(dpre).T @ f2(h_in)
            g[k + ".bias"] = f2(dpre).sum(0)
            dh = dpre @ p[k + ".weight"]
        dsf = dh  # gradient wrt the head's input feature -> wrt the state token (the cost embedding is detached)
        if self.cat_cf:
            dsf = dsf[..., :E]
        if self.mul_cf:
            dsf = dsf * c["c_e"]
        g["cost_pred_head.weight"] = f2(dlogits).T @ f2(af)
        g["cost_pred_head.bias"] = f2(dlogits).sum(0)
        g["state_pred_head.weight"] = f2(dsp).T @ f2(af)
        g["state_pred_head.bias"] = f2(dsp).sum(0)
        daf = dlogits @ p["cost_pred_head.weight"] + dsp @ p["state_pred_head.weight"]
        dout = np.zeros((B, T, R, E), dt)
        dout[:, :, R - 2], dout[:, :, R - 1] = dsf, daf
        dout = dout.reshape(B, R * T, E)
        if self.prefix:
            dout = np.concatenate([np.zeros((B, 1, E), dt), dout], 1)
        dx, g["out_norm.weight"], g["out_norm.bias"] = layer_norm_bwd(dout, c["ln_out"], p["out_norm.weight"])
        # ---- blocks backward
        for l in range(self.NL - 1, -1, -1):
            pre, bc = f"blocks.{l}.", c["blocks"][l]
            dm = dx * drop.get(f"res2_{l}", one)  # x = x + drop(mlp(n2))
            g[pre + "mlp.2.weight"] = f2(dm).T @ f2(bc["h"])
            g[pre + "mlp.2.bias"] = f2(dm).sum(0)
            dh = dm @ p[pre + "mlp.2.weight"]
            dhpre = dh * gelu_grad(bc["hpre"])
            g[pre + "mlp.0.weight"] = f2(dhpre).T @ f2(bc["n2"])
            g[pre + "mlp.0.bias"] = f2(dhpre).sum(0)
            dn2 = dhpre @ p[pre + "mlp.0.weight"]
            d2, g[pre + "norm2.weight"], g[pre + "norm2.bias"] = layer_norm_bwd(dn2, bc["ln2"], p[pre + "norm2.weight"])
            dx = dx + d2
            datt = dx * drop.get(f"res1_{l}", one)  # x = x + drop(att)
            g[pre + "attention.out_proj.weight"] = f2(datt).T @ f2(bc["o"])
            g[pre + "attention.out_proj.bias"] = f2(datt).sum(0)
            do = (datt @ p[pre + "attention.out_proj.weight"]).reshape(B, S, H, d).transpose(0, 2, 1, 3)
            P, q, k, v = bc["P"], bc["q"], bc["k"], bc["v"]
            dP = (do @ v.transpose(0, 1, 3, 2)) * drop.get(f"attn{l}", one)
            dv = bc["Pd"].transpose(0, 1, 3, 2) @ do
            dS = P * (dP - (dP * P).sum(-1, keepdims=True))
            dq = dS @ k / math.sqrt(d)
            dk = dS.transpose(0, 1, 3, 2) @ q / math.sqrt(d)
            dqkv = np.concatenate([t.transpose(0, 2, 1, 3).reshape(B, S, E) for t in (dq, dk, dv)], -1)
            g[pre + "attention.in_proj_weight"] = f2(dqkv).T @ f2(bc["n1"])
            g[pre + "attention.in_proj_bias"] = f2(dqkv).sum(0)
            dn1 = dqkv @ p[pre + "attention.in_proj_weight"]
            d1, g[pre + "norm1.weight"], g[pre + "norm1.bias"] = layer_norm_bwd(dn1, bc["ln1"], p[pre + "norm1.weight"])
            dx = dx + d1
        # ---- embeddings backward
        dseq, g["emb_norm.weight"], g["emb_norm.bias"] = layer_norm_bwd(dx * drop.get("emb", one), c["ln_emb"],
                                                                        p["emb_norm.weight"])
        if self.prefix:
            dpe, dseq = dseq[:, 0], dseq[:, 1:]
            g["prefix_emb.weight"] = (dpe * np.asarray(episode_cost, dt)[:, None]).sum(0)[:, None]
            g["prefix_emb.bias"] = dpe.sum(0)
        d4 = dseq.reshape(B, T, R, E)
        slot = 0
        if self.use_rew:
            dr = d4[:, :, slot]
            slot += 1
            g["return_emb.weight"] = (f2(dr) * returns.reshape(-1, 1)).sum(0)[:, None]
            g["return_emb.bias"] = f2(dr).sum(0)
        if self.use_cost:
            dc = d4[:, :, slot]
            slot += 1
            g["cost_emb.weight"] = (f2(dc) * c["ctg"].reshape(-1, 1)).sum(0)[:, None]
            g["cost_emb.bias"] = f2(dc).sum(0)
        ds, da = d4[:, :, R - 2], d4[:, :, R - 1]
        g["state_emb.weight"] = f2(ds).T @ f2(states)
        g["state_emb.bias"] = f2(ds).sum(0)
        g["action_emb.weight"] = f2(da).T @ f2(actions)
        g["action_emb.bias"] = f2(da).sum(0)
        if self.time_emb:
            dte = np.zeros_like(p["timestep_emb.weight"])
            np.add.at(dte, time_steps.reshape(-1), f2(d4.sum(2)))
            g["timestep_emb.weight"] = dte

        if grads_only:
            return g
        # ---- clip_grad_norm_ (cdt.py:398-399), AdamW with warm-up LR (cdt.py:321-330)
        if self.clip is not None:
            tot = math.sqrt(sum(float((np.asarray(v, np.float64) ** 2).sum()) for v in g.values()))
            coef = min(1.0, self.clip / (tot + 1e-6))
            if coef < 1.0:
                g = {k: v * dt(coef) for k, v in g.items()}
        lr_now = self.lr * min((self.steps + 1) / self.warmup, 1.0)
        self.opt.step(p, g, lr=lr_now)
        if self.stochastic:  # cdt.py:402-407
            gT = {"log_temperature": np.array(math.exp(self.log_temperature) * (ent - self.target_entropy))}
            pt = {"log_temperature": np.array(self.log_temperature, dtype=np.float64)}
            self.opt_T.step(pt, gT)
            self.log_temperature = float(pt["log_temperature"])
        self.steps += 1
        stats.update(all_loss=loss, act_loss=act_loss, cost_loss=cost_loss, cost_acc=acc, state_loss=state_loss,
                     train_lr=self.lr * min((self.steps + 1) / self.warmup, 1.0))
        return {k: float(v) for k, v in stats.items()}
