"""Batched, graph-captured policy rollouts for ``Trainer.evaluate()``.

The reference evaluates one episode at a time, one B=1 policy call and one host<->device round trip per env step
(``CPQTrainer.rollout`` cpq.py:330-347, ``BCQLTrainer.rollout`` bcql.py:323-340, ``BCTrainer.rollout``
bc.py:125-145).  Here the ``eval_episodes`` episodes are the ROWS of one policy batch: one env step of all
episodes = the policy's fused MLP forward(s) + its head kernel + one environment kernel (csrc/env.hip), captured
``_CHUNK`` env steps at a time in a hipGraph that is replayed ``episode_len / _CHUNK`` times; the host reads the per-episode totals once at the end.

Policies (same arithmetic as ``model.act``):
  BC    ``act_limit * tanh(pi(obs))``                                 bc.py:57-64  (multi-task: obs ++ cost_limit, :132-138)
  CPQ   ``max_action * tanh(mu(obs))`` (deterministic=True)           cpq.py:240-252, net.py:176-205 (BEAR-L too)
  DICE  ``tanh(mu(obs))``                                             coptidice.py:244-256
  BCQL  ``actor(obs, vae.decode(obs, z)), z = clamp(N(0,1), +-0.5)``  bcql.py:236-243, net.py:328-339
  CDT   windowed autoregression on the last seq_len steps             cdt.py:436-518 (``CDTBatchedRollout``)
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from ..common.net import actor_head_desc, net_desc_seq, vae_dec_desc
from . import glue as G
from .core import MlpRun, StepState, randn_fill, graph_capture

_Z_STREAM = 11  # Philox stream id of the BCQL decode noise drawn during evaluation
_CHUNK = 20     # env steps per captured graph: one replay costs ~25 us of launch latency, a step's kernels far less;
                # episodes that finish inside a chunk are frozen by the env kernel's done latch, so overshoot is a no-op


class BatchedRollout:
    """``eval_episodes`` episodes of one vector environment driven by ``model``'s policy on device."""

    def __init__(self, model, venv, kind: str, cost_scale: float = 1.0, extra_obs: Optional[float] = None,
                 seed: int = 0, z: Optional[torch.Tensor] = None, use_graph: bool = True):
        if kind not in ("bc", "cpq", "dice", "bcql"):
            raise ValueError(kind)
        m = self.model = model
        self.venv, self.kind, self.seed, self.use_graph = venv, kind, int(seed), use_graph
        dev = torch.device(m.device)
        E, od, ad = venv.E, venv.state_dim, m.action_dim
        f = dict(dtype=torch.float32, device=dev)
        in_dim = od + (1 if extra_obs is not None else 0)
        self.obs = torch.zeros(E, in_dim, **f)
        if extra_obs is not None:
            self.obs[:, od] = float(extra_obs)
        self.a = torch.zeros(E, ad, **f)
        self.env_c = venv.desc(m.episode_len, cost_scale)
        self.st = StepState(dev, ["x"])
        m.repack()
        if kind == "bc":
            if m.actor.pi[0].in_features != in_dim:
                raise ValueError(f"policy expects {m.actor.pi[0].in_features} inputs, the environment gives {in_dim}")
            self.r_pi = MlpRun(net_desc_seq([m.actor.pi], float(m.max_action)), E, False, dev)
        elif kind in ("cpq", "dice"):
            self.r_pi = MlpRun(actor_head_desc(m.actor), E, False, dev)
        else:
            Lz = m.latent_dim
            self.r_dec = MlpRun(vae_dec_desc(m.vae), E, False, dev)
            self.r_pi = MlpRun(net_desc_seq([m.actor.pi], 1.0), E, False, dev)
            self.z = torch.zeros(E, Lz, **f)
            self.z_fixed = z is not None
            if z is not None:  # fixed decode noise (parity tests); clamped like VAE.decode does (net.py:334-335)
                self.z.copy_(torch.as_tensor(z, **f).reshape(E, Lz))
                G.clamp_(self.z, -0.5, 0.5)
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    def body(self) -> None:
        m, E, ad = self.model, self.venv.E, self.model.action_dim
        if self.kind == "bc":
            act = self.r_pi.forward(self.obs)[0]
        elif self.kind in ("cpq", "dice"):  # COptiDICE.act does not scale by max_action (coptidice.py:252)
            head = self.r_pi.forward(self.obs)[0]
            G.gauss_head(head, None, E, ad, float(m.max_action) if self.kind == "cpq" else 1.0, a=self.a)
            act = self.a
        else:
            if not self.z_fixed:
                self.st.tick()
                randn_fill(self.z, self.seed, _Z_STREAM, self.st.ptr)
                G.clamp_(self.z, -0.5, 0.5)
            dec = self.r_dec.forward(self.obs, self.z)[0]
            t = self.r_pi.forward(self.obs, dec)[0]
            G.bcq_perturb(dec, t, E, ad, float(m.actor.phi), float(m.max_action), self.a)
            act = self.a
        self.venv.step(self.env_c, act, self.obs)

    def _capture(self) -> None:
        snap = (self.venv.state.clone(), self.venv.acc.clone(), self.obs.clone(), self.st.state.clone(),
                self.st.host_step)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.body()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with graph_capture(gr):
            for _ in range(_CHUNK):
                self.body()
        torch.cuda.synchronize()
        self.venv.state.copy_(snap[0]); self.venv.acc.copy_(snap[1]); self.obs.copy_(snap[2])
        self.st.state.copy_(snap[3])
        self.st.host_step = snap[4]
        self.graph = gr

    @torch.no_grad()
    def run(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """All episodes to completion -> per-episode (return, cost * cost_scale, length) as numpy arrays."""
        self.venv.reset(self.obs)
        steps = min(self.model.episode_len, self.venv.episode_len or self.model.episode_len)
        if self.env_c.episode_len != steps:  # the env descriptor is a by-value launch argument of the captured graph
            self.env_c.episode_len, self.graph = steps, None
        if self.use_graph and self.graph is None:
            self._capture()
        if self.use_graph:
            for _ in range((steps + _CHUNK - 1) // _CHUNK):
                self.graph.replay()
        else:
            for _ in range(steps):
                self.body()
        acc = self.venv.acc.cpu().numpy()  # the one host sync of the rollout
        return acc[:, 0].astype(np.float64), acc[:, 1].astype(np.float64), acc[:, 2].astype(np.float64)


def evaluate_batched(trainer, kind: str, eval_episodes: int, cost_scale: float = 1.0,
                     extra_obs: Optional[float] = None):
    """``Trainer.evaluate`` on a ``VecSyntheticSafeEnv``: (mean return, mean cost * cost_scale, mean length).
    The rollout object (buffers + captured graph) is cached on the trainer; it reads the model's packed weights in
    place, so it stays valid across train steps."""
    venv = trainer.env
    if venv.E != eval_episodes:
        raise ValueError(f"the vector environment holds {venv.E} episodes, evaluate() was asked for {eval_episodes}")
    key = (id(venv), kind, float(cost_scale), extra_obs)
    ro = getattr(trainer, "_rollout", None)
    if ro is None or ro[0] != key:
        ro = (key, BatchedRollout(trainer.model, venv, kind, cost_scale, extra_obs,
                                  use_graph=getattr(trainer, "use_graph", True)))
        trainer._rollout = ro
    ret, cost, length = ro[1].run()
    return float(ret.mean()), float(cost.mean()), float(length.mean())


class CDTBatchedRollout:
    """``CDTTrainer.rollout`` (cdt.py:436-518) for E episodes at once.  The reference re-slices the last ``seq_len``
    steps of a full-history buffer every env step; here an inference ``CDTEngine`` with batch = E holds the window
    in its batch buffers (left-aligned while it fills, sliding afterwards; csrc/env.hip ``cdt_push_kernel``), so one
    env step of all episodes = transformer forward -> pick the mean action at the last filled position -> env
    kernel -> window push, ``_CHUNK`` steps per captured graph."""

    def __init__(self, model, venv, cost_scale: float = 1.0, cost_reverse: bool = False, use_graph: bool = True):
        from .cdt import CDTEngine
        m = self.model = model
        self.venv, self.use_graph = venv, use_graph
        self.cost_scale, self.cost_reverse = float(cost_scale), bool(cost_reverse)
        dev = torch.device(m.device)
        E = venv.E
        cfg = m._engine.cfg if m._engine is not None else dict(
            learning_rate=1e-4, weight_decay=1e-4, betas=(0.9, 0.999), clip_grad=0.25, lr_warmup_steps=1,
            loss_cost_weight=0.0, loss_state_weight=0.0, no_entropy=False)
        self.eng = CDTEngine(m, E, cfg, inference=True)
        f = dict(dtype=torch.float32, device=dev)
        self.obs = torch.zeros(E, venv.state_dim, **f)
        self.act = torch.zeros(E, m.action_dim, **f)
        self.step_out = torch.zeros(E, 2, **f)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.env_c = venv.desc(m.episode_len, 1.0)  # acc[:,1] = raw cost sum, as cdt.py:513 accumulates it
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    def _reset(self, target_return: float, target_cost: float) -> None:
        e = self.eng
        self.venv.reset(self.obs)
        for t in (e.states, e.actions, e.returns, e.ctg, e.mask, e.costs):
            t.zero_()
        e.time_steps.zero_()
        e.states[:, 0].copy_(self.obs)
        e.returns[:, 0] = float(target_return)
        e.ctg[:, 0] = float(target_cost)
        e.episode_cost.fill_(float(target_cost))  # the cost-prefix token's input (cdt.py:459,481)
        e.mask[:, 0] = 1.0
        self.cursor.zero_()

    def body(self) -> None:
        from .. import _lib as L
        from .core import cur_stream
        m, e, E, lib = self.model, self.eng, self.venv.E, L.load()
        e.forward(train=False)
        L.check(lib.osrl_cdt_rollout_pick(e.head.data_ptr(), e.head.shape[1], m.action_dim, E, m.seq_len,
                                          self.cursor.data_ptr(), float(m.max_action), self.act.data_ptr(),
                                          cur_stream()), "osrl_cdt_rollout_pick")
        self.venv.step(self.env_c, self.act, self.obs, self.step_out)
        L.check(lib.osrl_cdt_rollout_push(e.states.data_ptr(), e.actions.data_ptr(), e.returns.data_ptr(),
                                          e.ctg.data_ptr(), e.time_steps.data_ptr(), e.mask.data_ptr(), E, m.seq_len,
                                          m.state_dim, m.action_dim, self.act.data_ptr(), self.obs.data_ptr(),
                                          self.obs.stride(0), self.step_out.data_ptr(), self.cost_scale,
                                          int(self.cost_reverse), self.cursor.data_ptr(), int(self.env_c.episode_len),
                                          cur_stream()),
                "osrl_cdt_rollout_push")

    def _capture(self) -> None:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.body()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with graph_capture(gr):
            for _ in range(_CHUNK):
                self.body()
        torch.cuda.synchronize()
        self.graph = gr

    @torch.no_grad()
    def run(self, target_return: float, target_cost: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Per-episode (return, raw cost sum, length)."""
        m = self.model
        steps = min(m.episode_len, self.venv.episode_len or m.episode_len)
        if self.env_c.episode_len != steps:
            self.env_c.episode_len, self.graph = steps, None
        m.repack()
        if self.use_graph and self.graph is None:
            self._capture()  # runs the body on scratch state; _reset below starts the real rollout
        self._reset(target_return, target_cost)
        if self.use_graph:
            for _ in range((steps + _CHUNK - 1) // _CHUNK):
                self.graph.replay()
        else:
            for _ in range(steps):
                self.body()
        acc = self.venv.acc.cpu().numpy()
        return acc[:, 0].astype(np.float64), acc[:, 1].astype(np.float64), acc[:, 2].astype(np.float64)
