"""The COptiDICE train step as a static launch plan on MI355X.

Follows ``COptiDICE.update`` (osrl/algorithms/coptidice.py:135-232): optimal weights w* :122-131 -> (chi, tau)
upper-bound estimator :149-185 -> nu loss :188-194 -> lambda loss :197-201 -> weighted-likelihood policy
extraction :204-217.  The nu and chi ensembles read the stacked rows ``[obs; next_obs]`` (2B rows), so one saved
forward per network serves the s and the s' terms and one backward + dW pass serves both gradients.

Batch-global statistics (the softmax over the batch in the chi loss, every mean) are computed by single-workgroup
kernels.  Data parallel (``dist``): every 1/B is the global batch; each rank all-gathers its ``ell`` rows so that the
softmax statistics (D_kl, chi_loss, the weights) are the GLOBAL ones on every rank (the scalar leaves tau / lmbda then
step identically everywhere); collectives per step: the ell gather, [chi grads | weighted_c share], nu grads,
[actor grads | statistics].
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import _lib as L
from ..common.net import actor_head_desc, net_desc_seq
from .core import DwPlan, MlpRun, StepState, capture_step, cur_stream, load_into, check_plans_current

STAT_KEYS = ["loss/chi_loss", "loss/tau_loss", "loss/D_kl", "loss/Df", "loss/td_error", "loss/nu_loss",
             "loss/lmbda_loss", "loss/actor_loss", "loss/tau", "loss/lmbda"]
NOISE_KEYS = ["obs_eps", "act_eps"]
F_TYPES = {"chi2": 0, "softchi": 1, "kl": 2}  # include/osrl_amd.h OSRL_F_*


class COptiDICEEngine:
    def __init__(self, model, batch_size: int, rows_global: int = 0, seed: int = 0, dist=None):
        m = self.model = model
        B = self.B = int(batch_size)
        self.dist = dist
        self.seed = seed if dist is None else dist.rank_seed(seed)  # independent noise per rank
        self.rows_global = int(rows_global) if dist is not None else 0
        if dist is not None and self.rows_global != B * dist.world:
            raise ValueError("data parallel COptiDICE needs rows_global = batch_size * world (equal shards)")
        dev = torch.device(m.device)
        od, ad = m.state_dim, m.action_dim
        f = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f)  # noqa: E731
        self.st = StepState(dev, STAT_KEYS)
        self.x2 = z(2 * B, od)  # [obs; next_obs]
        self.obs, self.nobs = self.x2[:B], self.x2[B:]
        self.act, self.rew, self.cost, self.done, self.init = z(B, ad), z(B), z(B), z(B), z(B)
        tot = B * od + B * ad
        self.noise_flat = z((tot + 3) // 4 * 4)
        self.noise: Dict[str, torch.Tensor] = {"obs_eps": self.noise_flat[:B * od].view(B, od),
                                               "act_eps": self.noise_flat[B * od:tot].view(B, ad)}
        self.d_nu = net_desc_seq(list(m.nu_network.q_nets), 1.0)
        self.d_chi = net_desc_seq(list(m.chi_network.q_nets), 1.0)
        self.d_actor = actor_head_desc(m.actor)
        g = m.groups
        m.repack()
        nn_, nc = m.num_nu, m.num_chi
        self.r_nu = MlpRun(self.d_nu, 2 * B, True, dev)
        self.dnu = z(nn_, 2 * B, 1)
        self.r_nu.setup_backward(self.dnu)
        self.p_nu = DwPlan(g["nu_network"], self.r_nu.dw_entries(), 2 * B, dev)
        self.r_nu_b = MlpRun(self.d_nu, 2 * B, False, dev)  # second evaluation with the updated nu (policy extraction)
        self.use_chi = m.cost_ub_epsilon != 0
        if self.use_chi:
            self.r_chi = MlpRun(self.d_chi, 2 * B, True, dev)
            self.dchi = z(nc, 2 * B, 1)
            self.r_chi.setup_backward(self.dchi)
            self.p_chi = DwPlan(g["chi_network"], self.r_chi.dw_entries(), 2 * B, dev)
        self.e, self.w, self.w2, self.ell = z(B), z(B), z(B), z(B)
        self.work = z(4)
        self.obs_n, self.act_n = z(B, od), z(B, ad)
        self.r_actor = MlpRun(self.d_actor, B, True, dev)
        self.dhead = z(1, B, 2 * ad)
        self.r_actor.setup_backward(self.dhead)
        self.p_actor = DwPlan(g["actor"], self.r_actor.dw_entries(), B, dev)
        # every dW plan of this engine is built: the slab epochs they were built against are recorded NOW (not at the
        # first step), so an engine that is constructed directly, never stepped and then superseded is flagged stale
        from .core import slab_epochs
        self._slab_epochs = slab_epochs(self.model)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.replay = None

    def _adam(self, name: str, plan: DwPlan, extra=()) -> None:
        """dW, (data parallel: all-reduce of the flat gradient and of ``extra`` in one collective), Adam."""
        plan.launch()
        grp = self.model.groups[name]
        if self.dist is not None:
            self.dist.all_reduce_many_([self.dist.reduce_local(grp), *extra])
        grp.adam_step(self.model._lrs[name], self.st.ptr)

    def body(self, device_noise: bool) -> None:
        m, st, B, lib = self.model, self.st, self.B, L.load()
        od, ad = m.state_dim, m.action_dim
        nn_, nc, ft = m.num_nu, m.num_chi, F_TYPES[m.f_type]
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        s = cur_stream
        dp, rg = self.dist, self.rows_global
        share = 1.0 if dp is None else 1.0 / dp.world
        st.prologue(self.replay, (self.obs, self.nobs, self.act, self.rew, self.cost, self.done, self.init), self.noise_flat, self.seed, device_noise)
        leaves = m.scalar_leaves

        nu2 = self.r_nu.forward(self.x2)
        L.check(lib.osrl_dice_optimal_w(p(nu2), nn_, B, p(self.rew), p(self.cost), p(self.done), p(leaves),
                                        p(self.work), 0, float(m.alpha), float(m.gamma), ft, p(self.e), p(self.w), s()),
                "osrl_dice_optimal_w")
        chi2 = self.r_chi.forward(self.x2) if self.use_chi else None
        ell_all = None
        if dp is not None and self.use_chi:  # the softmax of the chi loss runs over the GLOBAL batch
            L.check(lib.osrl_dice_chi_ell(p(chi2), nc, B, p(self.w), p(self.cost), p(self.done), p(self.init),
                                          float(m.gamma), float(m.init_state_propotion), p(self.ell), s()),
                    "osrl_dice_chi_ell")
            ell_all = dp.all_gather_concat(self.ell)
        L.check(lib.osrl_dice_chi_step(p(chi2), nc, B, p(self.w), p(self.cost), p(self.done), p(self.init),
                                       float(m.gamma), float(m.init_state_propotion), float(m.cost_ub_epsilon),
                                       float(m.scalar_lr), st.ptr, p(leaves), p(self.work), p(self.ell),
                                       p(self.dchi) if self.use_chi else None, p(ell_all), rg,
                                       0 if dp is None else dp.rank * B, share, st.stat_ptr("loss/chi_loss"), s()),
                "osrl_dice_chi_step")
        wc = self.work[2:3]  # this rank's share of weighted_c -> global
        if self.use_chi:
            self.r_chi.backward_dz()
            self._adam("chi_network", self.p_chi, extra=(wc,) if dp is not None else ())
        elif dp is not None:
            dp.all_reduce_(wc)
        L.check(lib.osrl_dice_nu_step(p(nu2), nn_, B, p(self.e), p(self.w), p(self.done), p(self.init), ft,
                                      float(m.gamma), float(m.alpha), float(m.init_state_propotion), float(m.qc_thres),
                                      float(m.scalar_lr), rg, share, st.ptr, p(leaves), p(self.work), p(self.dnu),
                                      st.stat_ptr("loss/Df"), s()), "osrl_dice_nu_step")
        self.r_nu.backward_dz()
        self._adam("nu_network", self.p_nu)

        # 2. policy extraction: noisy observations / actions, w* re-evaluated with the updated nu network
        L.check(lib.osrl_dice_perturb(p(self.obs), p(self.noise["obs_eps"]), p(m.observations_std), B, od, 0.1,
                                      p(self.obs_n), s()), "osrl_dice_perturb")
        L.check(lib.osrl_dice_perturb(p(self.act), p(self.noise["act_eps"]), p(m.actions_std), B, ad, 0.1,
                                      p(self.act_n), s()), "osrl_dice_perturb")
        head = self.r_actor.forward(self.obs_n)[0]
        nu2b = self.r_nu_b.forward(self.x2)
        L.check(lib.osrl_dice_optimal_w(p(nu2b), nn_, B, p(self.rew), p(self.cost), p(self.done), p(leaves),
                                        p(self.work), 1, float(m.alpha), float(m.gamma), ft, None, p(self.w2), s()),
                "osrl_dice_optimal_w")
        L.check(lib.osrl_dice_actor_loss(p(head), p(self.act_n), p(self.w2), B, ad, rg, p(self.dhead),
                                         st.stat_ptr("loss/actor_loss"), s()), "osrl_dice_actor_loss")
        self.r_actor.backward_dz()
        self._adam("actor", self.p_actor, extra=(st.stats,) if dp is not None else ())

    def load_batch(self, observations, next_observations, actions, rewards, costs, done, is_init) -> None:
        load_into(((self.obs, observations), (self.nobs, next_observations), (self.act, actions),
                   (self.rew, rewards), (self.cost, costs), (self.done, done), (self.init, is_init)))

    def _snapshot(self):
        m = self.model
        snap = {"leaves": m.scalar_leaves.clone(), "state": self.st.state.clone(), "host": self.st.host_step,
                "stats": self.st.stats.clone(), "ring": self.st.ring.clone()}
        for n, g in m.groups.items():
            snap[n] = (g.p.clone(), g.m.clone(), g.v.clone())
        return snap

    def _restore(self, snap) -> None:
        m = self.model
        m.scalar_leaves.copy_(snap["leaves"])
        self.st.state.copy_(snap["state"]); self.st.stats.copy_(snap["stats"]); self.st.ring.copy_(snap["ring"])
        self.st.host_step = snap["host"]
        for n, g in m.groups.items():
            pp, mm, v = snap[n]
            g.p.copy_(pp); g.m.copy_(mm); g.v.copy_(v)
        m.repack()

    def capture(self) -> None:
        snap = self._snapshot()
        g, self._arena = capture_step(self.st.state.device, lambda: self.body(True), lambda: self.body(True))
        torch.cuda.synchronize()
        self._restore(snap)
        self.graph = g

    def attach_replay(self, store) -> None:
        """Sample minibatches on device from ``store`` (a ``ReplayStore(..., state_init=True)``) inside the step."""
        if store is not None and not store.state_init:
            raise ValueError("COptiDICE needs the is_init flag: build the ReplayStore with state_init=True")
        self.replay = store
        self.graph = None

    def step_replay(self, use_graph: bool = True) -> None:
        check_plans_current(self)
        assert self.replay is not None
        if use_graph and self.dist is None:
            if self.graph is None:
                self.capture()
            self.graph.replay()
            self.st.host_step += 1
        else:
            self.body(True)

    def step(self, observations, next_observations, actions, rewards, costs, done, is_init, noise=None,
             use_graph: bool = True) -> None:
        check_plans_current(self)
        if self.replay is not None:
            raise RuntimeError("a replay store is attached: call step_replay() (or attach_replay(None))")
        self.load_batch(observations, next_observations, actions, rewards, costs, done, is_init)
        if noise is not None:
            for k in NOISE_KEYS:
                self.noise[k].copy_(torch.as_tensor(noise[k]).reshape(self.noise[k].shape), non_blocking=True)
            self.body(False)
            return
        if use_graph and self.dist is None:
            if self.graph is None:
                self.capture()
            self.graph.replay()
            self.st.host_step += 1
        else:
            self.body(True)
