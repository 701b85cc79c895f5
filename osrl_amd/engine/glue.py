"""Thin typed wrappers over the glue kernels of libosrl_amd.so (csrc/glue.hip).

Arguments are torch CUDA tensors (or raw device addresses for ``stat``); every call is an
asynchronous launch on the current stream.  No arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .. import _lib as L
from . import plan as _plan
from .core import cur_stream


def _p(t) -> Optional[int]:
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()


def gauss_head(head, eps, rows, ad, max_action, a=None, tanh_u=None, logp=None):
    L.check(L.load().osrl_gauss_head(_p(head), _p(eps), rows, ad, max_action, _p(a), _p(tanh_u), _p(logp),
                                     cur_stream()), "osrl_gauss_head")


def gauss_head_bwd(head, eps, tanh_u, da_nets, n_nets, rows, ad, max_action, dhead):
    L.check(L.load().osrl_gauss_head_bwd(_p(head), _p(eps), _p(tanh_u), _p(da_nets), n_nets, rows, ad, max_action,
                                         _p(dhead), cur_stream()), "osrl_gauss_head_bwd")


def gauss_ood_sample(head, eps, n_samples, rows, ad, out):
    L.check(L.load().osrl_gauss_ood_sample(_p(head), _p(eps), n_samples, rows, ad, _p(out), cur_stream()),
            "osrl_gauss_ood_sample")


def vae_latent(head, eps, rows, Lz, z):
    L.check(L.load().osrl_vae_latent(_p(head), _p(eps), rows, Lz, _p(z), cur_stream()), "osrl_vae_latent")


def vae_latent_tail(eps, Lz, z) -> "L.TailT":
    """osrl_mlp_tail_t for MlpRun.forward(tail=...): vae_latent() fused behind the encoder's forward launch."""
    t = L.TailT()
    t.kind, t.L, t.eps, t.out = L.TAIL_VAE_LATENT, Lz, _p(eps), _p(z)
    return t


def vae_latent_bwd_tail(head, eps, Lz, beta, rows_global, dhead) -> "L.TailT":
    """osrl_mlp_tail_t for MlpRun.backward_dz(tail=...): vae_latent_bwd() fused behind the decoder's backward launch."""
    t = L.TailT()
    t.kind, t.L, t.eps, t.head, t.out = L.TAIL_VAE_LATENT_BWD, Lz, _p(eps), _p(head), _p(dhead)
    t.beta, t.rows_global = beta, rows_global
    return t


def vae_kl_tail(Lz, kl) -> "L.TailT":
    """osrl_mlp_tail_t for MlpRun.forward(tail=...): vae_kl_rows() on the encoder's output, by the forward launch itself
    (the 80-row N*B-row kernel) or as a launch behind it (any other kernel)."""
    t = L.TailT()
    t.kind, t.L, t.out = L.TAIL_VAE_KL, int(Lz), _p(kl)
    t._keep = kl
    return t


VAE_TAILS = _plan.knob("OSRL_VAE_TAILS", "1", "reparameterisation / its backward as tails of the MLP launches") == "1"  # 0: the reparameterisation and its backward as own launches


def gauss_tail(ad, max_action, eps=None, a=None, eps2=None, a2=None, tanh2=None, eps_ood=None, n_samples=0,
               sampled=None):
    """osrl_mlp_tail_t of kind GAUSS: the squashed-Gaussian head's action draws made by the actor trunk's own forward
    launch (== gauss_head / gauss_ood_sample on its output)."""
    t = L.TailT()
    t.kind, t.L, t.max_action = L.TAIL_GAUSS, int(ad), float(max_action)
    t.eps, t.out = _p(eps), _p(a)
    t.eps2, t.out2, t.tanh2 = _p(eps2), _p(a2), _p(tanh2)
    t.eps_ood, t.out_ood, t.n_samples = _p(eps_ood), _p(sampled), int(n_samples)
    t._keep = (eps, a, eps2, a2, tanh2, eps_ood, sampled)
    return t


def vae_encode(r_enc, obs, act, eps, Lz, z):
    """head = encoder(obs, act); z = mean + exp(clamp(log_std)) * eps   (net.py:319-331) -- ONE launch: the latent is
    produced by the encoder's forward launch from its LDS-resident output tile.  Returns head [rows, 2 Lz]."""
    if VAE_TAILS:
        return r_enc.forward(obs, act, tail=vae_latent_tail(eps, Lz, z))[0]
    head = r_enc.forward(obs, act)[0]
    vae_latent(head, eps, r_enc.rows, Lz, z)
    return head


def vae_decoder_backward(r_dec, head, eps, Lz, beta, rows_global, dhead):
    """Backward of the decoder down to dL/dz (its dX slice) and on through the reparameterisation + the KL term to
    dL/d(encoder head) -- ONE launch."""
    if VAE_TAILS:
        r_dec.backward_dz(tail=vae_latent_bwd_tail(head, eps, Lz, beta, rows_global, dhead))
        return
    r_dec.backward_dz()
    vae_latent_bwd(head, eps, r_dec.dx, r_dec.rows, Lz, beta, rows_global, dhead)


VAE_NS_AUTO = True  # False: the plan chooser never picks the all-CU VAE launches (the seeds-vs-loss-launches bit-equality test)


class VaeNs:
    """The VAE phase's forward + backward as all-CU layer launches (csrc/vae_ns.hip, ``osrl_vae_ns_forward`` /
    ``osrl_vae_ns_backward``): the same buffers the four fused launches fill -- the encoder's / decoder's saved
    activations, ``z``, every dZ the dW plan reads, the logged ``loss_vae`` -- so everything behind it (dW, Adam) is
    unchanged.  ``r_enc`` / ``r_dec``: the training-row MlpRuns with their backward already set up.
    ``VaeNs.build`` returns None where the library does not take the shape (hidden width % 80, <= 448, ...)."""

    def __init__(self, r_enc, r_dec, obs, act, eps, z, Lz, beta, rows_global, stat):
        import torch
        dev, rows, H = obs.device, r_enc.rows, r_enc.net.dims[1]
        v = self.c = L.VaeNsT()
        v.enc, v.dec = C.pointer(r_enc.net.c), C.pointer(r_dec.net.c)
        v.rows, v.od, v.ad, v.L = rows, obs.shape[1], act.shape[1], int(Lz)
        v.rows_global, v.beta = int(rows_global), float(beta)
        v.obs, v.act, v.eps = _p(obs), _p(act), _p(eps)
        v.enc_acts, v.dec_acts, v.z = r_enc.acts_c, r_dec.acts_c, _p(z)
        v.enc_g, v.dec_g = r_enc.grads_c, r_dec.grads_c
        f = dict(dtype=torch.float32, device=dev)
        self.P = torch.zeros(rows, H, **f)
        self.slabs = torch.zeros(3 * (H // 80) * rows * 32, **f)
        self.partials = torch.zeros(2 * ((rows + 47) // 48), **f)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        v.P, v.slabs, v.partials, v.counter, v.stat = _p(self.P), _p(self.slabs), _p(self.partials), \
            self.counter.data_ptr(), _p(stat)
        self._keep = (r_enc, r_dec, obs, act, eps, z)

    @staticmethod
    def build(r_enc, r_dec, obs, act, eps, z, Lz, beta, rows_global, stat):
        if r_enc.grads_c is None or r_dec.grads_c is None or r_enc.net.E != 1 or r_dec.net.E != 1 or r_enc.net.nl != 3:
            return None
        if r_enc.net.dims[1] % 80 or r_enc.net.dims[1] > 448:
            return None
        ns = VaeNs(r_enc, r_dec, obs, act, eps, z, Lz, beta, rows_global, stat)
        return ns if L.load().osrl_vae_ns_supported(C.byref(ns.c)) == 1 else None

    def forward(self) -> None:
        L.check(L.load().osrl_vae_ns_forward(C.byref(self.c), cur_stream()), "osrl_vae_ns_forward")

    def backward(self) -> None:
        L.check(L.load().osrl_vae_ns_backward(C.byref(self.c), cur_stream()), "osrl_vae_ns_backward")


def loss_ws(device):
    """Scratch of one call site of the grid loss kernels (osrl_vae_loss_ws / osrl_bcq_critic_loss_ws): zeroed once."""
    import torch
    return torch.zeros(L.LOSS_WS, dtype=torch.float32, device=device)


def vae_loss(u, act, head, rows, ad, Lz, beta, rows_global, du, stat, ws=None):
    """``ws`` (loss_ws()): the grid form -- for batches beyond ~2048 rows."""
    if ws is not None:
        L.check(L.load().osrl_vae_loss_ws(_p(u), _p(act), _p(head), rows, ad, Lz, beta, rows_global, _p(du), _p(stat),
                                          _p(ws), cur_stream()), "osrl_vae_loss_ws")
        return
    L.check(L.load().osrl_vae_loss(_p(u), _p(act), _p(head), rows, ad, Lz, beta, rows_global, _p(du), _p(stat),
                                   cur_stream()), "osrl_vae_loss")


def vae_latent_bwd(head, eps, dz, rows, Lz, beta, rows_global, dhead):
    L.check(L.load().osrl_vae_latent_bwd(_p(head), _p(eps), _p(dz), rows, Lz, beta, rows_global, _p(dhead),
                                         cur_stream()), "osrl_vae_latent_bwd")


def vae_kl_rows(head, rows, Lz, kl):
    L.check(L.load().osrl_vae_kl_rows(_p(head), rows, Lz, _p(kl), cur_stream()), "osrl_vae_kl_rows")


def quantile(x, n, q, out):
    L.check(L.load().osrl_quantile(_p(x), n, q, _p(out), cur_stream()), "osrl_quantile")


def quantile_ws(x, n, q, ws, out):
    """Multi-workgroup select for large n; ``ws``: int32/uint32 tensor of L.QUANTILE_WS zeros (re-zeroed by the call)."""
    L.check(L.load().osrl_quantile_ws(_p(x), n, q, ws.data_ptr(), _p(out), cur_stream()), "osrl_quantile_ws")


def cpq_critic_loss(q_old, n_q_old, qc_old, n_qc_old, q, n_q, rew, done, rows, gamma, q_thres, rows_global, dq,
                    stat):
    L.check(L.load().osrl_cpq_critic_loss(_p(q_old), n_q_old, _p(qc_old), n_qc_old, _p(q), n_q, _p(rew), _p(done),
                                          rows, gamma, q_thres, rows_global, _p(dq), _p(stat), cur_stream()),
            "osrl_cpq_critic_loss")


def cpq_ood_mean(qc_sampled, n_qc_old, kl, quant, n_samples, rows, rows_global, out):
    L.check(L.load().osrl_cpq_ood_mean(_p(qc_sampled), n_qc_old, _p(kl), _p(quant), n_samples, rows, rows_global,
                                       _p(out), cur_stream()), "osrl_cpq_ood_mean")


def cpq_ood_stat(qc_sampled, n_qc_old, kl, q, n_samples, rows, rows_global, quant_out, out):
    L.check(L.load().osrl_cpq_ood_stat(_p(qc_sampled), n_qc_old, _p(kl), q, n_samples, rows, rows_global,
                                       _p(quant_out), _p(out), cur_stream()), "osrl_cpq_ood_stat")


def cpq_ood_select(kl, n, q, quant_out, row_list, count, quantile_in=None):
    """Quantile of kl[0..n) (or ``quantile_in``) and the ascending list of the indices that reach it (cpq.py:183-184)."""
    L.check(L.load().osrl_cpq_ood_select(_p(kl), None if quantile_in is None else _p(quantile_in), q, n, _p(quant_out),
                                         row_list.data_ptr(), count.data_ptr(), cur_stream()), "osrl_cpq_ood_select")


def cpq_ood_sum(qc_sel, n_qc, cap, count, scale, out):
    L.check(L.load().osrl_cpq_ood_sum(_p(qc_sel), n_qc, cap, count.data_ptr(), scale, _p(out), cur_stream()),
            "osrl_cpq_ood_sum")


def cpq_cost_loss(qc_old_next, n_qc_old, qc, n_qc, ood_mean, cost, rows, gamma, qc_thres, alpha_lr, rows_global,
                  stat_share, log_alpha, dq, stat):
    L.check(L.load().osrl_cpq_cost_loss(_p(qc_old_next), n_qc_old, _p(qc), n_qc, _p(ood_mean), _p(cost), rows, gamma,
                                        qc_thres, alpha_lr, rows_global, stat_share, _p(log_alpha), _p(dq), _p(stat),
                                        cur_stream()), "osrl_cpq_cost_loss")


def cpq_alpha_step(ood_mean, qc_thres, alpha_lr, stat_share, log_alpha, stat):
    L.check(L.load().osrl_cpq_alpha_step(_p(ood_mean), qc_thres, alpha_lr, stat_share, _p(log_alpha), _p(stat),
                                         cur_stream()), "osrl_cpq_alpha_step")


def cpq_cost_loss_ood(qc_sampled, n_qc_s, kl, quant, n_samples, qc_old_next, n_qc_old, qc, n_qc, ood_mean, cost, rows,
                      gamma, qc_thres, alpha_lr, log_alpha, dq, stat):
    L.check(L.load().osrl_cpq_cost_loss_ood(_p(qc_sampled), n_qc_s, _p(kl), _p(quant), n_samples, _p(qc_old_next),
                                            n_qc_old, _p(qc), n_qc, _p(ood_mean), _p(cost), rows, gamma, qc_thres,
                                            alpha_lr, _p(log_alpha), _p(dq), _p(stat), cur_stream()),
            "osrl_cpq_cost_loss_ood")


def cpq_actor_loss(q, n_q, qc, n_qc, rows, q_thres, rows_global, dq, stat):
    L.check(L.load().osrl_cpq_actor_loss(_p(q), n_q, _p(qc), n_qc, rows, q_thres, rows_global, _p(dq), _p(stat),
                                         cur_stream()), "osrl_cpq_actor_loss")


def mse_loss(u, target, n, n_global, du, stat):
    L.check(L.load().osrl_mse_loss(_p(u), _p(target), n, n_global, _p(du), _p(stat), cur_stream()),
            "osrl_mse_loss")


def bcq_perturb(dec, t, rows, ad, phi, max_action, a):
    L.check(L.load().osrl_bcq_perturb(_p(dec), _p(t), rows, ad, phi, max_action, _p(a), cur_stream()),
            "osrl_bcq_perturb")


def bcq_perturb_bwd(dec, t, da_nets, n_nets, rows, ad, phi, max_action, dt):
    L.check(L.load().osrl_bcq_perturb_bwd(_p(dec), _p(t), _p(da_nets), n_nets, rows, ad, phi, max_action, _p(dt),
                                          cur_stream()), "osrl_bcq_perturb_bwd")


def bcq_critic_loss(q_t, n1, n2, n_samples, q_on, n_on, base, done, rows, gamma, lmbda, rows_global, dq, stat, ws=None):
    if ws is not None:
        L.check(L.load().osrl_bcq_critic_loss_ws(_p(q_t), n1, n2, n_samples, _p(q_on), n_on, _p(base), _p(done), rows,
                                                 gamma, lmbda, rows_global, _p(dq), _p(stat), _p(ws), cur_stream()),
                "osrl_bcq_critic_loss_ws")
        return
    L.check(L.load().osrl_bcq_critic_loss(_p(q_t), n1, n2, n_samples, _p(q_on), n_on, _p(base), _p(done), rows,
                                          gamma, lmbda, rows_global, _p(dq), _p(stat), cur_stream()),
            "osrl_bcq_critic_loss")


def bcq_actor_sums(q, nq1, nq2, qc, nc1, nc2, rows, rows_global, out):
    L.check(L.load().osrl_bcq_actor_sums(_p(q), nq1, nq2, _p(qc), nc1, nc2, rows, rows_global, _p(out), cur_stream()),
            "osrl_bcq_actor_sums")


def bcq_actor_loss(q, nq1, nq2, qc, nc1, nc2, rows, qc_thres, KP, KI, KD, rows_global, pid, dq, dqc, stat,
                   global_means=None, stat_share=1.0):
    L.check(L.load().osrl_bcq_actor_loss(_p(q), nq1, nq2, _p(qc), nc1, nc2, rows, qc_thres, KP, KI, KD, rows_global,
                                         _p(global_means), stat_share, _p(pid), _p(dq), _p(dqc), _p(stat),
                                         cur_stream()), "osrl_bcq_actor_loss")


def clamp_(x, lo, hi):
    L.check(L.load().osrl_clamp(_p(x), x.numel(), lo, hi, cur_stream()), "osrl_clamp")


# ---- loss seeds (osrl_mlp_seed_t): the backward launch computes dL/d(output) itself --------------------------------
SEEDS = _plan.knob("OSRL_SEEDS", "1", "backward launches compute the gradient they start from", operator=True) == "1"  # 0: the loss kernels as launches of their own (A/B, bit-equality tests)


class SeedStat:
    """Scratch of one seeded call site: the per-tile partials of its logged statistic + the arrival counter."""

    def __init__(self, device, n_nets: int, rows: int):
        import torch
        self.partials = torch.zeros(2 * max(n_nets, 1) * ((rows + 15) // 16), dtype=torch.float32, device=device)
        self.counter = torch.zeros(1, dtype=torch.int32, device=device)


def _seed(kind, rows, rows_global, ws: Optional[SeedStat], stat) -> "L.SeedT":
    s = L.SeedT()
    s.kind, s.rows_global = kind, int(rows_global)
    if ws is not None:
        s.partials, s.counter, s.stat = ws.partials.data_ptr(), ws.counter.data_ptr(), _p(stat)
    s._keep = [ws]
    return s


def _inv(rows, rows_global):
    import numpy as np
    return np.float32(1.0) / np.float32(rows_global if rows_global > 0 else rows)


def seed_vae(act, head, rows, ad, Lz, beta, rows_global, ws, stat) -> "L.SeedT":
    """== vae_loss(u, act, head, ...): du = 2 (u - act) inv / ad; stat = rec inv / ad + beta * KL inv / L."""
    import numpy as np
    s = _seed(L.SEED_MSE, rows, rows_global, ws, stat)
    inv = _inv(rows, rows_global)
    s.x0, s.kl_head, s.kl_L = _p(act), _p(head), int(Lz)
    s.scale = s.stat_scale = float(inv / np.float32(ad))
    s.stat_scale2, s.kl_beta = float(inv / np.float32(Lz)), float(beta)
    s._keep += [act, head]
    return s


def seed_cpq_critic(q_old, n_q_old, qc_old, n_qc_old, rew, done, rows, gamma, q_thres, rows_global, ws, stat):
    """== cpq_critic_loss(...) for the online critics of the launch."""
    s = _seed(L.SEED_CPQ_CRITIC, rows, rows_global, ws, stat)
    s.a, s.n_a, s.b, s.n_b, s.x0, s.x1 = _p(q_old), n_q_old, _p(qc_old), n_qc_old, _p(rew), _p(done)
    s.gamma, s.thres = float(gamma), float(q_thres)
    s.scale = s.stat_scale = float(_inv(rows, rows_global))
    s._keep += [q_old, qc_old, rew, done]
    return s


def seed_cpq_cost(qc_old_next, n_qc_old, cost, rows, gamma, rows_global, ws, stat):
    """== the MSE part of cpq_cost_loss(...) (the dual step stays with cpq_alpha_step)."""
    s = _seed(L.SEED_CPQ_COST, rows, rows_global, ws, stat)
    s.a, s.n_a, s.x0 = _p(qc_old_next), n_qc_old, _p(cost)
    s.gamma = float(gamma)
    s.scale = s.stat_scale = float(_inv(rows, rows_global))
    s._keep += [qc_old_next, cost]
    return s


def seed_cpq_actor(q, n_q, qc, n_qc, rows, q_thres, rows_global, ws, stat):
    """== cpq_actor_loss(...): the launch's nets are the n_q critics whose outputs are ``q``."""
    s = _seed(L.SEED_CPQ_ACTOR, rows, rows_global, ws, stat)
    s.a, s.n_a, s.b, s.n_b = _p(q), n_q, _p(qc), n_qc
    s.thres = float(q_thres)
    s.scale = s.stat_scale = float(_inv(rows, rows_global))
    s._keep += [q, qc]
    return s


def seed_gauss_head(eps, tanh_u, da_nets, n_nets, rows, max_action):
    """== gauss_head_bwd(head, eps, tanh_u, da_nets, ...): head is the launch's own saved output."""
    s = _seed(L.SEED_GAUSS_HEAD, rows, 0, None, None)
    s.eps, s.tanh_u, s.a, s.n_a = _p(eps), _p(tanh_u), _p(da_nets), n_nets
    s.max_action = float(max_action)
    s._keep += [eps, tanh_u, da_nets]
    return s


def seed_bcq_critic(q_t, n1, n2, n_samples, base, done, rows, gamma, lmbda, rows_global, ws, stat):
    """== bcq_critic_loss(q_t, n1, n2, n_samples, q_on, ..., base, done, ...) for the online nets of the launch."""
    s = _seed(L.SEED_BCQ_CRITIC, rows, rows_global, ws, stat)
    s.a, s.n_a, s.n_b, s.n_samples, s.x0, s.x1 = _p(q_t), n1, n2, int(n_samples), _p(base), _p(done)
    s.gamma, s.thres = float(gamma), float(lmbda)
    s.scale = s.stat_scale = float(_inv(rows, rows_global))
    s._keep += [q_t, base, done]
    return s
