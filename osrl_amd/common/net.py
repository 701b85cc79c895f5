"""Network containers with the reference's names and ``state_dict`` layout
(osrl/common/net.py of liuzuxin/OSRL), re-hosted for MI355X.

These modules hold parameters only -- as VIEWS into flat HBM optimizer groups
(engine/core.py FlatGroup) -- plus thin ``forward`` methods that call the fused HIP
MLP through ``osrl_amd.ops`` (a ``torch.autograd.Function`` over libosrl_amd.so).
There is no CPU path, and no aten arithmetic on the train-step path of BC / CPQ / BCQ-Lag / BEAR-Lag / COptiDICE / CDT's
default configuration; the one aten op in this file is the element-wise ``torch.minimum`` over an ensemble's outputs in
``predict()`` (an inference convenience the step plans never call: they take the minimum inside their loss kernels).
Exception (engine/cdt.py): CDT's cost-feature (add / mul / cat) and cost-prefix constructor variants form the head's
input feature and compact the prefix token's row with 3-6 small aten element-wise / copy launches inside the captured
step (the maths, detach and mul-backward of cdt.py:243-250 included); the default training configuration has none.

Construction mirrors the reference's module/parameter creation ORDER (nn.Linear default
init draws from the global torch RNG), so ``torch.manual_seed(s)`` followed by building a
model yields the same initial weights as the reference under the same seed.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from ..engine.core import FlatGroup, LayerRef, NetDesc

_ACT_NAME = {nn.ReLU: "relu", nn.Tanh: "tanh", nn.Identity: "id"}


def mlp(sizes: Sequence[int], activation, output_activation=nn.Identity) -> nn.Sequential:
    """Linear/activation stack with the reference's Sequential indexing (net.py:12-30):
    Linear layers sit at even indices, so keys are ``{0,2,4,...}.{weight,bias}``."""
    mods: List[nn.Module] = []
    n = len(sizes) - 1
    for j in range(n):
        mods.append(nn.Linear(int(sizes[j]), int(sizes[j + 1])))
        mods.append((activation if j < n - 1 else output_activation)())
    return nn.Sequential(*mods)


def seq_linears(seq: nn.Sequential) -> List[nn.Linear]:
    return [m for m in seq if isinstance(m, nn.Linear)]


def seq_acts(seq: nn.Sequential) -> List[str]:
    return [_ACT_NAME[type(m)] for m in seq if not isinstance(m, nn.Linear)]


class MLPActor(nn.Module):
    """net.py:65-85: ``act_limit * tanh(mlp(obs))``."""

    def __init__(self, obs_dim, act_dim, hidden_sizes, activation, act_limit=1):
        super().__init__()
        self.pi = mlp([obs_dim] + list(hidden_sizes) + [act_dim], activation, nn.Tanh)
        self.act_limit = act_limit

    def forward(self, obs):
        from .. import ops
        return ops.mlp_apply(net_desc_seq([self.pi], self.act_limit), obs)[0]


class MLPGaussianPerturbationActor(nn.Module):
    """net.py:33-62: ``clamp(act + phi*act_limit*tanh(mlp([obs,act])), +-act_limit)``."""

    def __init__(self, obs_dim, act_dim, hidden_sizes, activation, phi=0.05, act_limit=1):
        super().__init__()
        self.pi = mlp([obs_dim + act_dim] + list(hidden_sizes) + [act_dim], activation, nn.Tanh)
        self.act_limit = act_limit
        self.phi = phi

    def forward(self, obs, act):
        from .. import ops
        t = ops.mlp_apply(net_desc_seq([self.pi], 1.0), obs, act)[0]
        return ops.bcq_perturb(act, t, self.phi, self.act_limit)


class SquashedGaussianMLPActor(nn.Module):
    """net.py:152-205: ReLU trunk + mu / log_std heads, tanh-squashed Gaussian."""

    def __init__(self, obs_dim, act_dim, hidden_sizes, activation):
        super().__init__()
        self.net = mlp([obs_dim] + list(hidden_sizes), activation, activation)
        self.mu_layer = nn.Linear(hidden_sizes[-1], act_dim)
        self.log_std_layer = nn.Linear(hidden_sizes[-1], act_dim)

    def forward(self, obs, deterministic=False, with_logprob=True, eps=None):
        """Returns (tanh(u), logp).  ``eps`` = explicit standard-normal noise [rows, act_dim]
        (drawn on device when omitted and not deterministic)."""
        from .. import ops
        head = ops.mlp_apply(actor_head_desc(self), obs)[0]
        return ops.gauss_head(head, eps, deterministic, with_logprob)


class EnsembleQCritic(nn.Module):
    """net.py:208-242."""

    def __init__(self, obs_dim, act_dim, hidden_sizes, activation, num_q=2):
        super().__init__()
        assert num_q >= 1, "num_q param should be greater than 1"
        self.q_nets = nn.ModuleList(
            [mlp([obs_dim + act_dim] + list(hidden_sizes) + [1], nn.ReLU) for _ in range(num_q)])

    def forward(self, obs, act=None):
        from .. import ops
        y = ops.mlp_apply(net_desc_seq(list(self.q_nets), 1.0), obs, act)
        return [y[e, :, 0] for e in range(len(self.q_nets))]

    def predict(self, obs, act):
        """net.py:235-238: (element-wise minimum over the ensemble, list of the members' values).  The minimum is a
        view-free reduction over <= 8 [rows] vectors: torch.minimum on device tensors (memory plumbing, like the
        stacking the reference does); the networks themselves run on the fused HIP kernels."""
        q_list = self.forward(obs, act)
        qmin = q_list[0]
        for q in q_list[1:]:
            qmin = torch.minimum(qmin, q)
        return qmin, q_list


class EnsembleDoubleQCritic(nn.Module):
    """net.py:245-287."""

    def __init__(self, obs_dim, act_dim, hidden_sizes, activation, num_q=2):
        super().__init__()
        assert num_q >= 1, "num_q param should be greater than 1"
        mk = lambda: mlp([obs_dim + act_dim] + list(hidden_sizes) + [1], nn.ReLU)  # noqa: E731
        self.q1_nets = nn.ModuleList([mk() for _ in range(num_q)])
        self.q2_nets = nn.ModuleList([mk() for _ in range(num_q)])

    def all_nets(self) -> List[nn.Sequential]:
        return list(self.q1_nets) + list(self.q2_nets)

    def forward(self, obs, act):
        from .. import ops
        y = ops.mlp_apply(net_desc_seq(self.all_nets(), 1.0), obs, act)
        n = len(self.q1_nets)
        return [y[e, :, 0] for e in range(n)], [y[n + e, :, 0] for e in range(n)]

    def predict(self, obs, act):
        q1, q2 = self.forward(obs, act)
        return torch.min(torch.vstack(q1), 0).values, torch.min(torch.vstack(q2), 0).values, q1, q2


class VAE(nn.Module):
    """net.py:290-339 (decode_multiple is BEARL-only and out of scope)."""

    def __init__(self, obs_dim, act_dim, hidden_size, latent_dim, act_lim, device="cuda"):
        super().__init__()
        self.e1 = nn.Linear(obs_dim + act_dim, hidden_size)
        self.e2 = nn.Linear(hidden_size, hidden_size)
        self.mean = nn.Linear(hidden_size, latent_dim)
        self.log_std = nn.Linear(hidden_size, latent_dim)
        self.d1 = nn.Linear(obs_dim + latent_dim, hidden_size)
        self.d2 = nn.Linear(hidden_size, hidden_size)
        self.d3 = nn.Linear(hidden_size, act_dim)
        self.act_lim = act_lim
        self.latent_dim = latent_dim
        self.device = device

    def decode(self, obs, z=None):
        from .. import ops
        if z is None:
            z = ops.randn((obs.shape[0], self.latent_dim), obs.device).clamp_(-0.5, 0.5)
        return ops.mlp_apply(vae_dec_desc(self), obs, z)[0]


class TransformerBlock(nn.Module):
    """Parameter container of the pre-LN GPT block (net.py:391-441); the arithmetic lives in engine/cdt.py."""

    def __init__(self, seq_len: int, embedding_dim: int, num_heads: int, attention_dropout: float,
                 residual_dropout: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.drop = nn.Dropout(residual_dropout)
        self.attention = nn.MultiheadAttention(embedding_dim, num_heads, attention_dropout, batch_first=True)
        self.mlp = nn.Sequential(nn.Linear(embedding_dim, 4 * embedding_dim), nn.GELU(),
                                 nn.Linear(4 * embedding_dim, embedding_dim), nn.Dropout(residual_dropout))
        # True = not allowed to attend (same buffer as the reference, net.py:417-418; part of the state_dict)
        self.register_buffer("causal_mask", ~torch.tril(torch.ones(seq_len, seq_len)).to(bool))
        self.seq_len = seq_len


class DiagGaussianActor(nn.Module):
    """Parameter container of the diagonal-Gaussian head (net.py:509-533)."""

    def __init__(self, hidden_dim, act_dim, log_std_bounds=(-5.0, 2.0)):
        super().__init__()
        self.mu = nn.Linear(hidden_dim, act_dim)
        self.log_std = nn.Linear(hidden_dim, act_dim)
        self.log_std_bounds = list(log_std_bounds)
        for m in (self.mu, self.log_std):  # same RNG consumption as the reference's orthogonal init
            nn.init.orthogonal_(m.weight.data)
            m.bias.data.fill_(0.0)


# --------------------------------------------------------------------------- #
# NetDesc builders (pointers into the modules' flat-group storage; set up by bind_group)
# --------------------------------------------------------------------------- #
def _ref(lin: nn.Linear) -> LayerRef:
    if not hasattr(lin, "_osrl"):
        raise RuntimeError("module parameters are not bound to a FlatGroup (model not materialised)")
    grp, wkey, bkey, tgt = lin._osrl
    return LayerRef(lin.weight.data, lin.bias.data, grp, wkey, bkey, tgt, [lin.weight], [lin.bias])


def net_desc_seq(seqs: Sequence[nn.Sequential], out_scale: float) -> NetDesc:
    return NetDesc([[_ref(l) for l in seq_linears(s)] for s in seqs], seq_acts(seqs[0]), out_scale)


def _packed(first: nn.Linear, second: nn.Linear) -> LayerRef:
    """[2k, H] weight / [2k] bias spanning two ADJACENT Linear layers (plan_group lays them out so)."""
    k, H = first.weight.shape
    w0, w1, b0, b1 = first.weight.data, second.weight.data, first.bias.data, second.bias.data
    if w1.data_ptr() != w0.data_ptr() + 4 * k * H or b1.data_ptr() != b0.data_ptr() + 4 * k:
        raise RuntimeError("packed head layers are not adjacent in HBM -- model was not materialised")
    grp, hw, hb, tgt = first._osrl_head
    return LayerRef(torch.as_strided(w0, (2 * k, H), (H, 1)), torch.as_strided(b0, (2 * k,), (1,)), grp, hw, hb, tgt,
                    [first.weight, second.weight], [first.bias, second.bias])


def actor_head_desc(actor: SquashedGaussianMLPActor) -> NetDesc:
    lins = seq_linears(actor.net)
    return NetDesc([[_ref(l) for l in lins] + [_packed(actor.mu_layer, actor.log_std_layer)]],
                   ["relu"] * len(lins) + ["id"], 1.0)


def vae_enc_desc(vae: VAE) -> NetDesc:
    return NetDesc([[_ref(vae.e1), _ref(vae.e2), _packed(vae.mean, vae.log_std)]], ["relu", "relu", "id"], 1.0)


def vae_dec_raw_desc(vae: VAE) -> NetDesc:
    """The decoder WITHOUT its tanh: ``d3(relu(d2(relu(d1(.)))))`` -- the second return value of
    ``VAE.decode_multiple`` (net.py:342-353), which BEAR-L's MMD term consumes."""
    return NetDesc([[_ref(vae.d1), _ref(vae.d2), _ref(vae.d3)]], ["relu", "relu", "id"], 1.0)


def vae_dec_desc(vae: VAE) -> NetDesc:
    return NetDesc([[_ref(vae.d1), _ref(vae.d2), _ref(vae.d3)]], ["relu", "relu", "tanh"], float(vae.act_lim))


# --------------------------------------------------------------------------- #
# materialisation: move a module's parameters into a FlatGroup (views)
# --------------------------------------------------------------------------- #
PACKED_PAIRS = (("mu_layer", "log_std_layer"), ("mean", "log_std"), ("mu", "log_std"))


def plan_group(group: FlatGroup, prefix: str, module: nn.Module) -> None:
    """Register every parameter of ``module`` (keys ``prefix.<name>``) in ``group``; the
    (mu|log_std) head pairs are laid out adjacently and aliased as ``<scope>.head.{weight,bias}``.
    Every Linear weight (or head alias) is marked for packing."""
    named: Dict[str, torch.Tensor] = dict(module.named_parameters())
    done = set()
    for name, p in named.items():
        if name in done:
            continue
        scope, _, leaf = name.rpartition(".")
        owner, _, mod = scope.rpartition(".")
        own = owner + "." if owner else ""
        pair = next((pr for pr in PACKED_PAIRS if mod == pr[0]), None)
        if pair is not None and own + pair[1] + ".weight" in named:
            first, second = own + pair[0], own + pair[1]
            w0, w1 = named[first + ".weight"], named[second + ".weight"]
            group.add(prefix + "." + first + ".weight", w0.shape)
            group.add(prefix + "." + second + ".weight", w1.shape, align=False)
            group.add(prefix + "." + first + ".bias", named[first + ".bias"].shape)
            group.add(prefix + "." + second + ".bias", named[second + ".bias"].shape, align=False)
            hk = prefix + "." + own + "head"
            group.alias(hk + ".weight", prefix + "." + first + ".weight", (2 * w0.shape[0], w0.shape[1]))
            group.alias(hk + ".bias", prefix + "." + first + ".bias", (2 * w0.shape[0],))
            group.mark_weight(hk + ".weight")
            done.update({first + ".weight", first + ".bias", second + ".weight", second + ".bias"})
        else:
            group.add(prefix + "." + name, p.shape)
            if leaf in ("weight", "in_proj_weight") and p.dim() == 2 and "timestep_emb" not in name:
                group.mark_weight(prefix + "." + name)
            done.add(name)


def bind_group(group: FlatGroup, prefix: str, module: nn.Module, target: Optional[nn.Module] = None) -> None:
    """Copy ``module``'s (CPU-initialised) parameters into the group and re-point ``.data`` at the
    flat views; ``target`` (a deepcopy) is bound to the Polyak target buffer the same way.  Every
    nn.Linear is tagged with its (group, weight key, bias key, is_target) for the NetDesc builders."""
    def bind(mod: nn.Module, is_tgt: bool):
        view = group.tgt_view if is_tgt else group.view
        with torch.no_grad():
            for name, p in mod.named_parameters():
                v = view(prefix + "." + name)
                v.copy_(p.data)
                p.data = v
                if is_tgt:
                    p.requires_grad_(False)
        for mname, sub in mod.named_modules():
            if isinstance(sub, nn.Linear):
                full = prefix + ("." + mname if mname else "")
                sub._osrl = (group, full + ".weight", full + ".bias", is_tgt)
                scope, _, leaf = mname.rpartition(".")
                if any(leaf == pr[0] for pr in PACKED_PAIRS):
                    hk = prefix + "." + (scope + "." if scope else "") + "head"
                    if hk + ".weight" in group.layout:
                        sub._osrl_head = (group, hk + ".weight", hk + ".bias", is_tgt)

    bind(module, False)
    if target is not None:
        bind(target, True)
