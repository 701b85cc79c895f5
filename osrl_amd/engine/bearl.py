"""The BEAR-Lagrangian train step as a static launch plan on MI355X.

Follows ``BEARLTrainer.train_one_step`` (osrl/algorithms/bearl.py:393-417): ``vae_loss`` :142-153 ->
``critic_loss`` :155-181 -> ``cost_critic_loss`` :183-209 -> ``actor_loss`` :211-275 (MMD support matching
:277-312, PID controller net.py:376-387, ``log_alpha`` dual step) -> ``sync_weight`` :329-335 (fused into each
group's Adam kernel).  The N*B-row target pipeline and the B*M-row actor pipeline never materialise the repeated
observations: the MLP kernels read ``obs[r / N]`` / ``obs[r / M]`` directly.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import _lib as L
from ..common.net import actor_head_desc, net_desc_seq, vae_dec_desc, vae_dec_raw_desc, vae_enc_desc
from . import glue as G
from .core import Branches, DwPlan, MlpRun, StepState, capture_step, concat_nets, cur_stream, load_into, check_plans_current

STAT_KEYS = ["loss/loss_vae", "loss/critic_loss", "loss/cost_critic_loss", "loss/actor_loss", "loss/mmd_loss",
             "loss/qc_penalty", "loss/lagrangian", "loss/alpha_value"]
NOISE_KEYS = ["eps_vae", "eps_c", "eps_cc", "z_mmd", "eps_pi"]
KERNELS = {"gaussian": 0, "laplacian": 1}  # include/osrl_amd.h OSRL_MMD_*


class BEARLEngine:
    def __init__(self, model, batch_size: int, rows_global: int = 0, seed: int = 0, dist=None):
        m = self.model = model
        B = self.B = int(batch_size)
        self.rows_global, self.dist = int(rows_global), dist
        self.seed = seed if dist is None else dist.rank_seed(seed)  # independent noise per rank
        dev = torch.device(m.device)
        od, ad, Lz, N, M = m.state_dim, m.action_dim, m.latent_dim, m.sample_action_num, m.num_samples_mmd_match
        nq, nqc = m.num_q, m.num_qc
        if 2 * nq + 2 * nqc > L.MAX_NETS:
            raise ValueError(f"2*num_q + 2*num_qc = {2 * nq + 2 * nqc} > {L.MAX_NETS} nets per fused launch")
        if M > 64 or ad > 16:
            raise ValueError("the MMD kernel supports num_samples_mmd_match <= 64 and action_dim <= 16")
        f = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f)  # noqa: E731
        self.st = StepState(dev, STAT_KEYS)
        self.obs, self.nobs, self.act = z(B, od), z(B, od), z(B, ad)
        self.rew, self.cost, self.done = z(B), z(B), z(B)
        shapes = {"eps_vae": (B, Lz), "eps_c": (N * B, ad), "eps_cc": (N * B, ad), "z_mmd": (B * M, Lz),
                  "eps_pi": (B * M, ad)}
        tot = sum(int(torch.Size(s).numel()) for s in shapes.values())
        self.noise_flat = z((tot + 3) // 4 * 4)
        self.noise: Dict[str, torch.Tensor] = {}
        o = 0
        for k in NOISE_KEYS:
            n = int(torch.Size(shapes[k]).numel())
            self.noise[k] = self.noise_flat[o:o + n].view(shapes[k])
            o += n

        twin = lambda mod: net_desc_seq(mod.all_nets(), 1.0)  # noqa: E731
        self.d_actor, self.d_actor_old = actor_head_desc(m.actor), actor_head_desc(m.actor_old)
        self.d_critic, self.d_cost = twin(m.critic), twin(m.cost_critic)
        self.d_critic_old, self.d_cost_old = twin(m.critic_old), twin(m.cost_critic_old)
        self.d_enc, self.d_dec, self.d_dec_raw = vae_enc_desc(m.vae), vae_dec_desc(m.vae), vae_dec_raw_desc(m.vae)
        g = m.groups
        m.repack()

        # vae phase
        self.r_enc, self.r_dec = MlpRun(self.d_enc, B, True, dev), MlpRun(self.d_dec, B, True, dev)
        self.z, self.du, self.dhead_enc = z(B, Lz), z(1, B, ad), z(1, B, 2 * Lz)
        self.r_dec.setup_backward(self.du, dx_cols=(od, Lz))
        self.r_enc.setup_backward(self.dhead_enc)
        self.p_vae = DwPlan(g["vae"], self.r_enc.dw_entries() + self.r_dec.dw_entries(), B, dev)

        # target pipelines (critic on the capture stream, cost critic on the side branch: two buffer sets)
        NB = N * B
        self.r_aold = [MlpRun(self.d_actor_old, NB, False, dev) for _ in range(2)]
        self.a_t = [z(NB, ad) for _ in range(2)]
        self.r_qold_t = MlpRun(self.d_critic_old, NB, False, dev)
        self.r_qcold_t = MlpRun(self.d_cost_old, NB, False, dev)
        self.r_critic = MlpRun(self.d_critic, B, True, dev)
        self.dq = z(2 * nq, B, 1)
        self.r_critic.setup_backward(self.dq)
        self.p_critic = DwPlan(g["critic"], self.r_critic.dw_entries(), B, dev)
        self.r_cost = MlpRun(self.d_cost, B, True, dev)
        self.dqc = z(2 * nqc, B, 1)
        self.r_cost.setup_backward(self.dqc)
        self.p_cost = DwPlan(g["cost_critic"], self.r_cost.dw_entries(), B, dev)

        # actor phase: B*M rows through the raw decoder and the actor; B rows (sample 0) through both critics
        BM = B * M
        self.r_dec_raw = MlpRun(self.d_dec_raw, BM, False, dev)
        self.r_actor = MlpRun(self.d_actor, BM, True, dev)
        self.mmd, self.du_mmd, self.tanh_u, self.a0 = z(B), z(BM, ad), z(BM, ad), z(B, ad)
        self.r_pi_q = MlpRun(concat_nets(self.d_critic, self.d_cost), B, True, dev)
        self.dq_pi = z(2 * nq + 2 * nqc, B, 1)
        self.r_pi_q.setup_backward(self.dq_pi, need_dz=False, dx_cols=(od, ad))
        self.dhead = z(1, BM, 2 * ad)
        self.coef, self.pi_means = z(4), z(4)
        self.r_actor.setup_backward(self.dhead)
        self.p_actor = DwPlan(g["actor"], self.r_actor.dw_entries(), BM, dev)
        # every dW plan of this engine is built: the slab epochs they were built against are recorded NOW (not at the
        # first step), so an engine that is constructed directly, never stepped and then superseded is flagged stale
        from .core import slab_epochs
        self._slab_epochs = slab_epochs(self.model)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.replay = None

    def _optim(self, name: str, plan: DwPlan, tau: float) -> None:
        plan.launch()
        self._update(name, tau)

    def _update(self, name: str, tau: float) -> None:
        grp = self.model.groups[name]
        if self.dist is not None:
            self.dist.allreduce_group(grp)
        grp.adam_step(self.model._lrs[name], self.st.ptr, tau=tau)

    def _targets(self, eps_key: str, r_q: MlpRun, which: int) -> torch.Tensor:
        """bearl.py:158-164: Q_old(obs', tanh(actor_old sample)) on the N*B repeated rows; the sampled action is NOT
        scaled by max_action (``self.actor_old(...)`` is called directly, not ``_actor_forward``)."""
        m, N, NB = self.model, self.model.sample_action_num, self.model.sample_action_num * self.B
        head = self.r_aold[which].forward(self.nobs, map0=L.MAP_DIV, div0=N)[0]
        G.gauss_head(head, self.noise[eps_key], NB, m.action_dim, 1.0, a=self.a_t[which])
        return r_q.forward(self.nobs, self.a_t[which], map0=L.MAP_DIV, div0=N)

    def _seeds(self):
        """The loss seeds of this engine's backward launches (glue.seed_*), built once."""
        if getattr(self, "_seed_cache", 0) == 0:
            m, B, dev = self.model, self.B, self.obs.device
            nq, nqc, N, rg = m.num_q, m.num_qc, m.sample_action_num, self.rows_global
            self._seed_cache = None
            if G.SEEDS and G.VAE_TAILS:
                self._seed_cache = {
                    "vae": G.seed_vae(self.act, self.r_enc.y[0], B, m.action_dim, m.latent_dim, m.beta, rg,
                                      G.SeedStat(dev, 1, B), self.st.stat_ptr("loss/loss_vae")),
                    "critic": G.seed_bcq_critic(self.r_qold_t.y, nq, nq, N, self.rew, self.done, B, m.gamma, m.lmbda, rg,
                                                G.SeedStat(dev, 2 * nq, B), self.st.stat_ptr("loss/critic_loss")),
                    "cost": G.seed_bcq_critic(self.r_qcold_t.y, nqc, nqc, N, self.cost, None, B, m.gamma, m.lmbda, rg,
                                              G.SeedStat(dev, 2 * nqc, B), self.st.stat_ptr("loss/cost_critic_loss")),
                }
        return self._seed_cache

    def body(self, device_noise: bool, par: Optional[Branches] = None) -> None:
        """``par`` (graph capture): cost_critic_loss reads only actor_old and cost_critic_old -- nothing the critic
        phase writes -- so it runs on a side branch beside critic_loss, followed there by the head of actor_loss
        (raw VAE decodes, actor samples, MMD), which needs the updated VAE and the not-yet-updated actor only."""
        par = par or Branches(False)
        m, st, nz, B, lib = self.model, self.st, self.noise, self.B, L.load()
        od, ad, Lz, N, M = m.state_dim, m.action_dim, m.latent_dim, m.sample_action_num, m.num_samples_mmd_match
        nq, nqc, rg = m.num_q, m.num_qc, self.rows_global
        st.prologue(self.replay, (self.obs, self.nobs, self.act, self.rew, self.cost, self.done), self.noise_flat, self.seed, device_noise)
        G.clamp_(nz["z_mmd"], -0.5, 0.5)  # decode_multiple clamps its latent draw (net.py:343-346)

        head = G.vae_encode(self.r_enc, self.obs, self.act, nz["eps_vae"], Lz, self.z)
        u = self.r_dec.forward(self.obs, self.z)[0]
        sd = self._seeds()
        if sd is not None:  # (round 4) the backward launches compute the gradient they start from, as in engine/cpq.py
            self.r_dec.backward_dz(tail=G.vae_latent_bwd_tail(head, nz["eps_vae"], Lz, m.beta, rg, self.dhead_enc),
                                   seed=sd["vae"])
        else:
            G.vae_loss(u, self.act, head, B, ad, Lz, m.beta, rg, self.du, st.stat_ptr("loss/loss_vae"))
            G.vae_decoder_backward(self.r_dec, head, nz["eps_vae"], Lz, m.beta, rg, self.dhead_enc)
        self.r_enc.backward_dz()
        self._optim("vae", self.p_vae, 0.0)

        par.fork(0)
        q_t = self._targets("eps_c", self.r_qold_t, 0)
        q = self.r_critic.forward(self.obs, self.act)
        if sd is not None:
            self.r_critic.backward_dz(seed=sd["critic"])
        else:
            G.bcq_critic_loss(q_t, nq, nq, N, q, 2 * nq, self.rew, self.done, B, m.gamma, m.lmbda, rg, self.dq,
                              st.stat_ptr("loss/critic_loss"))  # same lambda-mix / max-over-N backup as BCQ-L
            self.r_critic.backward_dz()
        if self.dist is None:
            self._optim("critic", self.p_critic, m.tau)
        else:  # reduced together with the cost critic's gradient after the join: one collective instead of two
            self.p_critic.launch()

        with par.on(0):
            qc_t = self._targets("eps_cc", self.r_qcold_t, 1)
            qc = self.r_cost.forward(self.obs, self.act)
            if sd is not None:
                self.r_cost.backward_dz(seed=sd["cost"])
            else:
                G.bcq_critic_loss(qc_t, nqc, nqc, N, qc, 2 * nqc, self.cost, None, B, m.gamma, m.lmbda, rg, self.dqc,
                                  st.stat_ptr("loss/cost_critic_loss"))
                self.r_cost.backward_dz()
            self.p_cost.launch()
            # head of actor_loss (bearl.py:219-232): needs the updated VAE and the not-yet-updated actor only
            raw = self.r_dec_raw.forward(self.obs, nz["z_mmd"], map0=L.MAP_DIV, div0=M)[0]
            ahead = self.r_actor.forward(self.obs, map0=L.MAP_DIV, div0=M)[0]
            L.check(lib.osrl_bear_mmd(raw.data_ptr(), ahead.data_ptr(), nz["eps_pi"].data_ptr(), B, M, ad,
                                      float(m.mmd_sigma), KERNELS[m.kernel], self.mmd.data_ptr(),
                                      self.du_mmd.data_ptr(), self.tanh_u.data_ptr(), self.a0.data_ptr(),
                                      cur_stream()), "osrl_bear_mmd")
        par.join(0)
        if self.dist is None:
            self._update("cost_critic", m.tau)
        else:
            gc, gcc = m.groups["critic"], m.groups["cost_critic"]
            self.dist.all_reduce_many_([self.dist.reduce_local(gc), self.dist.reduce_local(gcc)])
            gc.adam_step(m._lrs["critic"], st.ptr, tau=m.tau)
            gcc.adam_step(m._lrs["cost_critic"], st.ptr, tau=m.tau)

        y = self.r_pi_q.forward(self.obs, self.a0)
        yq, yqc = y[:2 * nq], y[2 * nq:]
        means, share = None, 1.0
        if self.dist is not None:  # PID and the dual step act on GLOBAL means
            L.check(lib.osrl_bear_actor_sums(yq.data_ptr(), nq, nq, yqc.data_ptr(), nqc, nqc, self.mmd.data_ptr(), B,
                                             rg, self.pi_means.data_ptr(), cur_stream()), "osrl_bear_actor_sums")
            self.dist.all_reduce_(self.pi_means)
            means, share = self.pi_means, 1.0 / self.dist.world
        L.check(lib.osrl_bear_actor_loss(yq.data_ptr(), nq, nq, yqc.data_ptr(), nqc, nqc, self.mmd.data_ptr(), B,
                                         float(m.qc_thres), float(m.KP), float(m.KI), float(m.KD),
                                         float(m.target_mmd_thresh), float(m.alpha_lr),
                                         int(m.start_update_policy_step), rg,
                                         None if means is None else means.data_ptr(), share, st.ptr,
                                         m.pid_state.data_ptr(), m.log_alpha.data_ptr(), self.dq_pi[:2 * nq].data_ptr(),
                                         self.dq_pi[2 * nq:].data_ptr(), self.coef.data_ptr(),
                                         st.stat_ptr("loss/actor_loss"), cur_stream()), "osrl_bear_actor_loss")
        self.r_pi_q.backward_dz()
        L.check(lib.osrl_bear_head_bwd(ahead.data_ptr(), nz["eps_pi"].data_ptr(), self.tanh_u.data_ptr(),
                                       self.du_mmd.data_ptr(), self.coef.data_ptr(), self.r_pi_q.dx.data_ptr(),
                                       2 * nq + 2 * nqc, B, M, ad, self.dhead.data_ptr(), cur_stream()),
                "osrl_bear_head_bwd")
        self.r_actor.backward_dz()
        if self.dist is None:
            self._optim("actor", self.p_actor, m.tau)
        else:  # actor gradient and the per-rank partial statistics in one collective
            self.p_actor.launch()
            ga = m.groups["actor"]
            self.dist.all_reduce_many_([self.dist.reduce_local(ga), st.stats])
            ga.adam_step(m._lrs["actor"], st.ptr, tau=m.tau)

    def load_batch(self, observations, next_observations, actions, rewards, costs, done) -> None:
        load_into(((self.obs, observations), (self.nobs, next_observations), (self.act, actions),
                   (self.rew, rewards), (self.cost, costs), (self.done, done)))

    def _snapshot(self):
        m = self.model
        snap = {"pid": m.pid_state.clone(), "la": m.log_alpha.clone(), "state": self.st.state.clone(),
                "host": self.st.host_step, "stats": self.st.stats.clone(), "ring": self.st.ring.clone()}
        for n, g in m.groups.items():
            snap[n] = (g.p.clone(), g.m.clone(), g.v.clone(), None if g.tgt is None else g.tgt.clone())
        return snap

    def _restore(self, snap) -> None:
        m = self.model
        m.pid_state.copy_(snap["pid"]); m.log_alpha.copy_(snap["la"])
        self.st.state.copy_(snap["state"]); self.st.stats.copy_(snap["stats"]); self.st.ring.copy_(snap["ring"])
        self.st.host_step = snap["host"]
        for n, g in m.groups.items():
            p, mm, v, t = snap[n]
            g.p.copy_(p); g.m.copy_(mm); g.v.copy_(v)
            if t is not None:
                g.tgt.copy_(t)
        m.repack()

    def capture(self) -> None:
        snap = self._snapshot()
        par = Branches(True, 1)
        g, self._arena = capture_step(self.st.state.device, lambda: self.body(True), lambda: self.body(True, par))
        self._par = par  # keep the side stream alive with the graph
        torch.cuda.synchronize()
        self._restore(snap)
        self.graph = g

    def attach_replay(self, store) -> None:
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

    def step(self, observations, next_observations, actions, rewards, costs, done, noise=None,
             use_graph: bool = True) -> None:
        check_plans_current(self)
        if self.replay is not None:
            raise RuntimeError("a replay store is attached: call step_replay() (or attach_replay(None))")
        self.load_batch(observations, next_observations, actions, rewards, costs, done)
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
