"""Tensor-level ops over libosrl_amd.so for code outside the captured step plan
(``model.act``, module ``forward`` methods, user losses).

``mlp_apply`` is a ``torch.autograd.Function`` whose forward AND backward are the fused HIP
kernels of csrc/mlp.hip (forward, backward-dz with dX, split-K dW).  Nothing here falls back to
aten arithmetic; without the library (or without a HIP device) every call raises.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib as L
from .engine import glue as G
from .engine.core import DwPlan, FlatGroup, MlpRun, NetDesc, randn_fill


def _chk(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("osrl_amd ops need HIP device tensors (no CPU fallback)")
    return x.contiguous().float()


def randn(shape, device, seed: int = 0, stream_id: int = 7) -> torch.Tensor:
    """Standard-normal noise from the on-device Philox generator (csrc/rng.hip)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _RandnCounter.n += 1
    randn_fill(out, seed + 0x9E3779B97F4A7C15 * _RandnCounter.n % (1 << 63), stream_id, None)
    return out


class _RandnCounter:
    n = 0


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, desc: NetDesc, x0: torch.Tensor, x1: Optional[torch.Tensor], *params):
        rows = x0.shape[0]
        need_grad = any(ctx.needs_input_grad)  # (grad mode is off inside Function.forward)
        for g in desc.groups():  # weights may have been edited since the last pack: refresh (cheap)
            g.repack()
        run = MlpRun(desc, rows, need_grad, x0.device)
        run.forward(x0, x1)
        ctx.run, ctx.desc, ctx.has_x1 = run, desc, x1 is not None
        ctx.d0 = x0.shape[1]
        return run.y

    @staticmethod
    def backward(ctx, dy):
        run, desc = ctx.run, ctx.desc
        dy = dy.contiguous()
        din = desc.dims[0]
        run.setup_backward(dy, need_dz=True, dx_cols=(0, din))
        run.backward_dz()
        # weight gradients through a temporary flat slab with this net's own layout
        grp = FlatGroup("tmp", dy.device)
        entries = []
        for e in range(desc.E):
            for l in range(desc.nl):
                wk, bk = f"{e}.{l}.w", f"{e}.{l}.b"
                grp.add(wk, desc.nets[e][l].W.shape)
                grp.add(bk, desc.nets[e][l].b.shape)
                a = run.x if l == 0 else run.h[e][l - 1]
                entries.append((run.dz[e][l], a, wk, bk))
        grp.finalize()
        plan = DwPlan(grp, entries, run.rows, dy.device)
        plan.launch()
        grads = []
        for e in range(desc.E):
            for l in range(desc.nl):
                r = desc.nets[e][l]
                gw, gb = grp.grad_view(f"{e}.{l}.w"), grp.grad_view(f"{e}.{l}.b")
                # packed heads: split the stacked gradient back onto the individual nn.Parameters
                ws = r.wparams if r.wparams is not None else [r.W]
                bs = r.bparams if r.bparams is not None else [r.b]
                o = 0
                for w in ws:
                    grads.append(gw[o:o + w.shape[0]].contiguous())
                    o += w.shape[0]
                o = 0
                for bb in bs:
                    grads.append(gb[o:o + bb.shape[0]].contiguous())
                    o += bb.shape[0]
        dx = run.dx.sum(0)
        d0 = ctx.d0
        return (None, dx[:, :d0].contiguous(), dx[:, d0:].contiguous() if ctx.has_x1 else None, *grads)


def mlp_apply(desc: NetDesc, x0: torch.Tensor, x1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[E, rows, out] = fused MLP ensemble on cat(x0, x1).  Differentiable w.r.t. inputs and weights
    when the NetDesc was built from live ``nn.Parameter`` storage (pass the params for autograd)."""
    x0 = _chk(x0)
    x1 = None if x1 is None else _chk(x1)
    params = [t for net in desc.nets for r in net
              for t in ((r.wparams if r.wparams is not None else [r.W]) + (r.bparams if r.bparams is not None else [r.b]))]
    return _FusedMLP.apply(desc, x0, x1, *params)


def gauss_head(head: torch.Tensor, eps: Optional[torch.Tensor], deterministic: bool, with_logprob: bool
               ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """SquashedGaussianMLPActor tail (net.py:176-205): returns (tanh(u), logp)."""
    head = _chk(head)
    rows, ad = head.shape[0], head.shape[1] // 2
    if not deterministic and eps is None:
        eps = randn((rows, ad), head.device)
    a = torch.empty(rows, ad, dtype=torch.float32, device=head.device)
    logp = torch.empty(rows, dtype=torch.float32, device=head.device) if with_logprob else None
    G.gauss_head(head, None if deterministic else _chk(eps), rows, ad, 1.0, a=a, logp=logp)
    return a, logp


def bcq_perturb(dec: torch.Tensor, t: torch.Tensor, phi: float, max_action: float) -> torch.Tensor:
    dec, t = _chk(dec), _chk(t)
    a = torch.empty_like(dec)
    G.bcq_perturb(dec, t, dec.shape[0], dec.shape[1], phi, max_action, a)
    return a


@torch.no_grad()
def cpq_act(model, obs: torch.Tensor, deterministic: bool):
    """CPQ._actor_forward (cpq.py:115-123) for inference: (max_action*tanh(u), logp)."""
    a, logp = model.actor(obs, deterministic, True)
    return a * model.max_action, logp
