"""Host side of the B = 1 (.. 4) ``act()`` latency path (csrc/act.hip, include/osrl_amd.h ``osrl_policy_*``).

The reference's evaluation loop calls ``model.act(obs)`` once per environment step (cpq.py:330-347 ->
cpq.py:240-252; bcql.py:236-243; bc.py:57-64): a host->device copy of one observation, a handful of aten kernels and
two ``.cpu().numpy()`` syncs.  ``FastPolicy`` keeps ONE pinned, device-mapped I/O block per model: ``act()`` writes
the observation into it through a numpy view, makes one C call (one kernel launch + a spin on the published sequence
number) and reads the action back through another numpy view -- no torch tensor is created on this path.

The kernel reads the packed forward weight copies of the flat optimizer groups -- the ones the fused optimizer kernel
and ``load_state_dict`` keep in step with the parameters -- and the canonical biases; the flat buffers never move, so a
``FastPolicy`` built once stays valid for the model's lifetime (in-place edits of parameters from outside the trainer
need ``model.repack()``, as for training).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib as L
from .core import NetDesc, cur_stream, require_cuda


def _gemv_net(dst: "L.GemvNetT", desc: NetDesc) -> None:
    if desc.E != 1:
        raise ValueError("the latency path runs one policy network per stage")
    dst.n_layers = desc.nl
    for i, v in enumerate(desc.dims):
        dst.dims[i] = v
    for i, a in enumerate(desc.acts):
        dst.acts[i] = a
    dst.out_scale = desc.out_scale
    for l, r in enumerate(desc.nets[0]):
        if r.target:
            raise ValueError("policies act with their online parameters")
        dst.Wf[l], dst.b[l] = r.wf_ptr, r.b.data_ptr()


class FastPolicy:
    """``kind``: "mlp" (BC), "gauss" (squashed-Gaussian actor), "bcq" (VAE decoder + perturbation actor)."""

    KINDS = {"mlp": L.POLICY_MLP, "gauss": L.POLICY_GAUSS, "bcq": L.POLICY_BCQ}

    def __init__(self, kind: str, device, obs_dim: int, act_dim: int, net0: NetDesc, max_action: float = 1.0,
                 net1: Optional[NetDesc] = None, latent_dim: int = 0, phi: float = 0.0, seed: int = 0):
        require_cuda(device)
        self.device = torch.device(device)
        d = L.PolicyT()
        d.kind, d.obs_dim, d.act_dim, d.latent_dim = self.KINDS[kind], obs_dim, act_dim, latent_dim
        d.max_action, d.phi = float(max_action), float(phi)
        _gemv_net(d.net[0], net0)
        if net1 is not None:
            _gemv_net(d.net[1], net1)
        self._keep = (net0, net1)  # the descriptors hold the parameter views alive
        self.kind, self.obs_dim, self.act_dim, self.seed = kind, obs_dim, act_dim, int(seed)
        self.noise_dim = {"mlp": 0, "gauss": act_dim, "bcq": latent_dim}[kind]
        lib = L.load()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(lib.osrl_policy_create(C.byref(d), C.byref(h)), "osrl_policy_create")
        self._h, self._lib = h, lib
        ptrs = [C.POINTER(C.c_float)() for _ in range(4)]
        L.check(lib.osrl_policy_io(h, *[C.byref(p) for p in ptrs]), "osrl_policy_io")
        R = L.POLICY_MAX_ROWS
        view = lambda p, shape: np.ctypeslib.as_array(p, shape=shape)  # noqa: E731  numpy views of PINNED memory
        self.obs = view(ptrs[0], (R, obs_dim))
        self.noise = view(ptrs[1], (R, max(self.noise_dim, 1)))
        self.act_out = view(ptrs[2], (R, act_dim))
        self.logp_out = view(ptrs[3], (R,))
        self._act1, self._obs1, self._lp1 = self.act_out[0], self.obs[0], self.logp_out[0:1].reshape(())
        self._fn = lib.osrl_policy_act
        self._gauss = kind == "gauss"
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)

    # The handle wraps a ctypes pointer to a pinned, device-mapped block: it cannot be copied or pickled.  A copied /
    # unpickled model simply has no fast policy yet and builds its own on its first act() (the models test `_fast is None`).
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def act1(self, obs, deterministic: bool = True):
        """The hot call of the episode loop: ONE observation [obs_dim], no explicit noise.  Everything a call does on
        the host: one numpy copy into pinned memory, one C call, one or two copies out."""
        if np.shape(obs) != self._obs1.shape:  # numpy would broadcast a scalar / length-1 observation silently
            raise ValueError(f"expected one observation of shape {self._obs1.shape}, got {np.shape(obs)}")
        self._obs1[:] = obs  # converts dtype
        st = self._raw_stream(self._dev_index) if self._raw_stream is not None else cur_stream()
        rc = self._fn(self._h, 1, 1 if deterministic else 0, 0, self.seed, st)
        if rc != 0:
            L.check(rc, "osrl_policy_act")
        return self._act1.copy(), (self._lp1.copy() if self._gauss else None)

    def act(self, obs, deterministic: bool = True, noise=None) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """``obs``: [obs_dim] or [rows <= 4, obs_dim].  Returns copies (action[, log-prob]) with the input's leading
        shape.  ``noise``: explicit standard-normal draws ([.., act_dim] eps for "gauss", [.., latent_dim] z for
        "bcq"); omitted = drawn in the kernel (Philox) when the policy is stochastic."""
        if noise is None and np.ndim(obs) == 1:
            return self.act1(obs, deterministic)
        o = np.asarray(obs, dtype=np.float32)
        single = o.ndim == 1
        rows = 1 if single else o.shape[0]
        if rows > L.POLICY_MAX_ROWS or o.shape[-1] != self.obs_dim:
            raise ValueError(f"expected [<= {L.POLICY_MAX_ROWS}, {self.obs_dim}] observations, got {o.shape}")
        if single:
            self._obs1[:] = o
        else:
            self.obs[:rows] = o
        host_noise = 0
        if noise is not None and self.noise_dim:
            self.noise[:rows, :self.noise_dim] = np.asarray(noise, np.float32).reshape(rows, self.noise_dim)
            host_noise = 1
        rc = self._fn(self._h, rows, 1 if deterministic else 0, host_noise, self.seed, cur_stream())
        if rc != 0:
            L.check(rc, "osrl_policy_act")
        if single:
            return self._act1.copy(), (self.logp_out[0].copy() if self.kind == "gauss" else None)
        return self.act_out[:rows].copy(), (self.logp_out[:rows].copy() if self.kind == "gauss" else None)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            self._lib.osrl_policy_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass
