#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (never copying it).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

Two stub modules (gymnasium, fsrl.utils) are injected so ``osrl.algorithms``
imports.  Inputs come from tests/cases.py (numpy RandomState, so
tests regenerate them bit-identically).  Gaussian noise is injected by patching
the samplers the reference calls (torch.randn / randn_like / Normal.rsample /
Normal.sample) to pop pre-drawn numpy tensors, in the reference's own draw order;
shapes are asserted, so a wrong order fails loudly.

Each fixture holds: per-step stats, final parameters after ``steps`` steps
(full tensors for small cases, a strided sample + per-tensor sums for wide ones),
scalar state (log_alpha / PID), optimizer moments of one tensor per optimizer,
and ``act()`` outputs on the batch observations.  torch/numpy versions are
recorded in the file.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from cases import (BEARL_CASES, CASES, CDT_CASES, COPTIDICE_CASES, dice_stds, cdt_drop_sites, hyper, make_batch, make_cdt_batch, make_cdt_drop,  # noqa: E402
                   make_cdt_params, make_noise, make_params, noise_shapes)

REF = os.environ.get("OSRL_REFERENCE", "/root/reference")


def _install_stubs():
    g = types.ModuleType("gymnasium")
    g.Env = object
    sys.modules["gymnasium"] = g
    f = types.ModuleType("fsrl")
    fu = types.ModuleType("fsrl.utils")

    class DummyLogger:
        def __init__(self, *a, **k):
            self.rows = []
            self.cur = {}

        def store(self, tab=None, **kw):
            self.cur.update(kw)

        def flush(self):
            self.rows.append(dict(self.cur))
            self.cur = {}

    fu.DummyLogger = DummyLogger
    fu.WandbLogger = DummyLogger
    f.utils = fu
    sys.modules["fsrl"] = f
    sys.modules["fsrl.utils"] = fu
    return DummyLogger


class NoiseQueue:
    """Replaces the reference's Gaussian samplers with a FIFO of numpy draws."""

    def __init__(self, torch):
        self.torch = torch
        self.q = []

    def push_step(self, case, step):
        nz = make_noise(case, step)
        for k, shape in noise_shapes(case):
            self.q.append((k, nz[k]))

    def pop(self, shape):
        k, v = self.q.pop(0)
        assert tuple(v.shape) == tuple(shape), (k, v.shape, tuple(shape))
        return self.torch.from_numpy(v.copy())

    def install(self):
        torch, q = self.torch, self
        import torch.distributions.normal as tdn

        torch.randn_like = lambda t, **kw: q.pop(t.shape)

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            return q.pop(size)

        torch.randn = randn
        tdn._standard_normal = lambda shape, dtype, device: q.pop(shape)
        # Normal.sample(): torch.normal(loc.expand(shape), scale.expand(shape))
        torch.normal = lambda mean, std, **kw: mean + std * q.pop(mean.shape)


def build(case, torch, algos, Logger):
    hp = hyper(case)
    lg = Logger()
    if case.algo == "bc":
        m = algos.BC(case.od, case.ad, case.max_action, case.hidden, case.episode_len)
        tr = algos.BCTrainer(m, None, lg, actor_lr=hp["actor_lr"])
    elif case.algo == "cpq":
        m = algos.CPQ(case.od, case.ad, case.max_action, case.hidden, case.hidden, case.vae_hidden,
                      case.N, hp["gamma"], hp["tau"], hp["beta"], case.num_q, case.num_qc,
                      hp["qc_scalar"], case.cost_limit, case.episode_len)
        tr = algos.CPQTrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["alpha_lr"], hp["vae_lr"])
    elif case.algo == "coptidice":
        ostd, astd = dice_stds(case)
        m = algos.COptiDICE(case.od, case.ad, case.max_action, hp["f_type"], hp["init_state_propotion"], ostd, astd,
                            case.hidden, case.hidden, hp["gamma"], hp["alpha"], hp["cost_ub_epsilon"], case.num_q,
                            case.num_qc, case.cost_limit, case.episode_len)
        tr = algos.COptiDICETrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["scalar_lr"])
    elif case.algo == "bearl":
        m = algos.BEARL(case.od, case.ad, case.max_action, case.hidden, case.hidden, case.vae_hidden, case.N,
                        hp["gamma"], hp["tau"], hp["beta"], hp["lmbda"], hp["mmd_sigma"], hp["target_mmd_thresh"],
                        hp["M"], list(hp["PID"]), hp["kernel"], case.num_q, case.num_qc, case.cost_limit,
                        case.episode_len, hp["start"])
        tr = algos.BEARLTrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["alpha_lr"], hp["vae_lr"])
    else:
        m = algos.BCQL(case.od, case.ad, case.max_action, case.hidden, case.hidden, case.vae_hidden,
                       case.N, hp["gamma"], hp["tau"], hp["phi"], hp["lmbda"], hp["beta"],
                       list(hp["PID"]), case.num_q, case.num_qc, case.cost_limit, case.episode_len)
        tr = algos.BCQLTrainer(m, None, lg, hp["actor_lr"], hp["critic_lr"], hp["vae_lr"])
    sd = {k: torch.from_numpy(v.copy()) for k, v in make_params(case).items()}
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m, tr, lg


def main(cases=None):
    cases = CASES if cases is None else cases
    Logger = _install_stubs()
    sys.path.insert(0, REF)
    import torch
    import osrl.algorithms as algos

    torch.set_num_threads(4)
    nq = NoiseQueue(torch)
    nq.install()
    for case in cases.values():
        m, tr, lg = build(case, torch, algos, Logger)
        b = {k: torch.from_numpy(v) for k, v in make_batch(case).items()}
        out = {}
        snap_steps = sorted({1, 3, case.steps} & set(range(1, case.steps + 1)))
        for s in range(case.steps):
            nq.push_step(case, s)
            if case.algo == "bc":
                tr.train_one_step(b["observations"], b["actions"])
            elif case.algo == "coptidice":
                tr.train_one_step([b[k] for k in ("observations", "next_observations", "actions", "rewards", "costs",
                                                  "done", "is_init")])
            else:
                tr.train_one_step(b["observations"], b["next_observations"], b["actions"],
                                  b["rewards"], b["costs"], b["done"])
            assert not nq.q, f"{case.name}: {len(nq.q)} noise tensors left unconsumed"
            lg.flush()
            if s + 1 in snap_steps:
                wide = case.name.endswith("_wide") or case.name == "bc_c1"
                for k, v in m.state_dict().items():
                    a = v.detach().numpy()
                    if wide:
                        out[f"p{s + 1}/sum/{k}"] = np.float64(a.astype(np.float64).sum())
                        out[f"p{s + 1}/abs/{k}"] = np.float64(np.abs(a.astype(np.float64)).sum())
                        out[f"p{s + 1}/smp/{k}"] = a.reshape(-1)[::97].copy()
                    else:
                        out[f"p{s + 1}/{k}"] = a.copy()
                if case.algo in ("cpq", "bearl"):
                    out[f"s{s + 1}/log_alpha"] = np.float64(m.log_alpha.item())
                if case.algo == "coptidice":
                    out[f"s{s + 1}/tau"] = np.float64(m.tau.item())
                    out[f"s{s + 1}/lmbda"] = np.float64(m.lmbda.item())
                if case.algo in ("bcql", "bearl"):
                    out[f"s{s + 1}/pid_error_old"] = np.float64(float(m.controller.error_old))
                    out[f"s{s + 1}/pid_error_integral"] = np.float64(float(m.controller.error_integral))
        keys = sorted(lg.rows[0].keys())
        out["stat_keys"] = np.array(keys)
        out["stats"] = np.array([[r[k] for k in keys] for r in lg.rows], dtype=np.float64)
        # one Adam moment pair per optimizer (first parameter of each)
        for oname in ("actor_optim", "critic_optim", "cost_critic_optim", "vae_optim", "nu_optim", "chi_optim"):
            opt = getattr(m, oname, None)
            if opt is None:
                continue
            p0 = opt.param_groups[0]["params"][0]
            st = opt.state[p0]
            if "exp_avg" not in st:  # an optimizer that never stepped (COptiDICE's chi with cost_ub_epsilon == 0)
                continue
            out[f"adam/{oname}/exp_avg"] = st["exp_avg"].numpy().copy()
            out[f"adam/{oname}/exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
            out[f"adam/{oname}/step"] = np.float64(float(st["step"]))
        # act() on the batch observations after training
        with torch.no_grad():
            if case.algo == "bc":
                out["act"] = m.actor(b["observations"]).numpy()
            elif case.algo in ("cpq", "bearl"):
                a, _ = m._actor_forward(b["observations"], True, True)
                out["act"] = a.numpy()
            elif case.algo == "coptidice":
                out["act"] = m.actor.forward(b["observations"], True, True)[0].numpy()
            else:
                z = np.random.RandomState(4000 + case.seed).randn(case.B, 2 * case.ad).astype(np.float32)
                dec = m.vae.decode(b["observations"], torch.from_numpy(z).clamp(-0.5, 0.5))
                out["act_z"] = z
                out["act"] = m.actor(b["observations"], dec).numpy()
        out["meta"] = np.array([f"torch={torch.__version__}", f"numpy={np.__version__}",
                                f"case={case}"])
        path = os.path.join(HERE, case.name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{case.name}: {os.path.getsize(path) / 1024:.1f} KB  stats[0]={dict(zip(keys, out['stats'][0]))}")


class DropQueue:
    """Feeds explicit keep-multipliers to every dropout draw of the reference, in its own call order.

    nn.Dropout -> F.dropout is replaced by ``x * mask``.  The attention-probability dropout happens inside the
    fused ``F.scaled_dot_product_attention`` that nn.MultiheadAttention calls (need_weights=False); that call is
    replaced by its textbook definition softmax(QK^T/sqrt(d) + mask) -> dropout -> @V written with torch ops
    (the dropout-free goldens pin the fused kernel itself).  Shapes are asserted.
    """

    def __init__(self, torch):
        self.torch, self.q = torch, []

    def push(self, c, step):
        m = make_cdt_drop(c, step)
        self.q += [(k, m[k]) for k, _ in cdt_drop_sites(c)]

    def pop(self, shape):
        k, v = self.q.pop(0)
        assert tuple(v.shape) == tuple(shape), (k, v.shape, tuple(shape))
        return self.torch.from_numpy(v)

    def install(self):
        torch, dq = self.torch, self
        import math
        F = torch.nn.functional

        def dropout(x, p=0.5, training=True, inplace=False):
            return x * dq.pop(x.shape) if (training and p > 0) else x

        def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, **kw):
            assert not is_causal
            s = q @ k.transpose(-2, -1) / math.sqrt(q.shape[-1])
            if attn_mask is not None:
                s = s.masked_fill(~attn_mask, float("-inf")) if attn_mask.dtype == torch.bool else s + attn_mask
            P = torch.softmax(s, -1)
            if dropout_p > 0:
                P = P * dq.pop(P.shape)
            return P @ v

        F.dropout = dropout
        F.scaled_dot_product_attention = sdpa


def main_cdt():
    """CDT goldens: fp32 mask; dropout 0 (the CDT class default, cdt.py:55-57) except the
    ``dropout`` cases, which inject explicit masks through DropQueue."""
    Logger = _install_stubs()
    sys.path.insert(0, REF)
    import torch
    import osrl.algorithms as algos

    torch.set_num_threads(4)
    dq = DropQueue(torch)
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--cdt-cases=")]
    todo = [c for c in CDT_CASES.values() if not only or c.name in only[0]]
    for c in sorted(todo, key=lambda c: c.dropout):  # patch the samplers only for the dropout cases
        if c.dropout > 0 and not dq.q and getattr(dq, "_on", None) is None:
            dq.install()
            dq._on = True
        m = algos.CDT(c.od, c.ad, 1.0, seq_len=c.T, episode_len=c.episode_len, embedding_dim=c.E,
                      num_layers=c.layers, num_heads=c.heads, attention_dropout=c.dropout,
                      residual_dropout=c.dropout, embedding_dropout=c.dropout, time_emb=c.time_emb, use_rew=c.use_rew,
                      use_cost=c.use_cost, cost_transform=c.cost_transform, add_cost_feat=c.add_cost_feat,
                      mul_cost_feat=c.mul_cost_feat, cat_cost_feat=c.cat_cost_feat, action_head_layers=c.head_layers,
                      cost_prefix=c.cost_prefix, stochastic=c.stochastic, init_temperature=0.1, target_entropy=-c.ad)
        lg = Logger()
        tr = algos.CDTTrainer(m, None, lg, learning_rate=c.lr, weight_decay=c.wd, betas=(0.9, 0.999),
                              clip_grad=c.clip, lr_warmup_steps=c.warmup, reward_scale=0.1, cost_scale=1.0,
                              loss_cost_weight=c.cost_w, loss_state_weight=c.state_w)
        sd = {k: torch.from_numpy(v.copy()) for k, v in make_cdt_params(c).items()}
        res = m.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys, res
        b = {k: torch.from_numpy(v) for k, v in make_cdt_batch(c).items()}
        out = {}
        for s in range(c.steps):
            if c.dropout > 0:
                dq.push(c, s)
            tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"],
                              b["mask"], b["episode_cost"], b["costs"])
            assert not dq.q, f"{c.name}: {len(dq.q)} dropout masks left unconsumed"
            lg.flush()
            if s + 1 in (1, c.steps):
                for k, v in m.state_dict().items():
                    if "causal_mask" in k:
                        continue
                    a = v.detach().numpy()
                    if c.name == "cdt_mid":
                        out[f"p{s + 1}/sum/{k}"] = np.float64(a.astype(np.float64).sum())
                        out[f"p{s + 1}/smp/{k}"] = a.reshape(-1)[::97].copy()
                    else:
                        out[f"p{s + 1}/{k}"] = a.copy()
                if c.stochastic:
                    out[f"s{s + 1}/log_temperature"] = np.float64(m.log_temperature.item())
        keys = sorted(lg.rows[0].keys())
        out["stat_keys"] = np.array(keys)
        out["stats"] = np.array([[r[k] for k in keys] for r in lg.rows], dtype=np.float64)
        with torch.no_grad():
            m.eval()
            ap, _, _ = m(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"],
                         ~b["mask"].to(torch.bool), b["episode_cost"])
            out["act"] = (ap.mean if c.stochastic else ap).numpy()
        out["meta"] = np.array([f"torch={torch.__version__}", f"numpy={np.__version__}", f"case={c}"])
        path = os.path.join(HERE, c.name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{c.name}: {os.path.getsize(path) / 1024:.1f} KB stats[0]={dict(zip(keys, out['stats'][0]))}")


if __name__ == "__main__":
    if "--bearl-only" in sys.argv:
        main(BEARL_CASES)
    elif "--coptidice-only" in sys.argv:
        main(COPTIDICE_CASES)
    else:
        if "--cdt-only" not in sys.argv:
            main()
            main(BEARL_CASES)
            main(COPTIDICE_CASES)
        main_cdt()
