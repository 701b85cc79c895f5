"""The BCQ-Lag train step as a static launch plan on MI355X.

Follows ``BCQLTrainer.train_one_step`` (osrl/algorithms/bcql.py:283-306): ``vae_loss`` :122-132 ->
``critic_loss`` :134-155 -> ``cost_critic_loss`` :157-179 -> ``actor_loss`` :181-216 (+ PID controller
net.py:376-387) -> ``sync_weight`` :228-234 (fused into each group's Adam kernel).
The N*B-row target pipeline (repeat_interleave -> vae.decode -> actor_old -> twin ensembles) never
materialises the repeated observations: the MLP kernels read ``next_obs[r / N]`` directly.
"""
from __future__ import annotations

import os

from typing import Dict, Optional

import torch

from .. import _lib as L
from ..common.net import net_desc_seq, vae_dec_desc, vae_enc_desc
from . import glue as G
from . import plan as P
from .core import Branches, DwPlan, MlpRun, StepState, capture_step, concat_nets, load_into, check_plans_current

STAT_KEYS = ["loss/loss_vae", "loss/critic_loss", "loss/cost_critic_loss", "loss/actor_loss", "loss/qc_penalty",
             "loss/lagrangian"]
NOISE_KEYS = ["eps_vae", "z_c", "z_cc", "z_actor"]


class BCQLEngine:
    def __init__(self, model, batch_size: int, rows_global: int = 0, seed: int = 0, dist=None):
        m = self.model = model
        B = self.B = int(batch_size)
        self.rows_global, self.dist = int(rows_global), dist
        self.seed = seed if dist is None else dist.rank_seed(seed)  # independent noise per rank
        dev = torch.device(m.device)
        od, ad, Lz, N = m.state_dim, m.action_dim, m.latent_dim, m.sample_action_num
        nq, nqc = m.num_q, m.num_qc
        if 2 * nq + 2 * nqc > L.MAX_NETS:
            raise ValueError(f"2*num_q + 2*num_qc = {2 * nq + 2 * nqc} > {L.MAX_NETS} nets per fused launch")
        f = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f)  # noqa: E731
        self.st = StepState(dev, STAT_KEYS)
        self.obs, self.nobs, self.act = z(B, od), z(B, od), z(B, ad)
        self.rew, self.cost, self.done = z(B), z(B), z(B)
        shapes = {"eps_vae": (B, Lz), "z_c": (N * B, Lz), "z_cc": (N * B, Lz), "z_actor": (B, Lz)}
        tot = sum(int(torch.Size(s).numel()) for s in shapes.values())
        self.noise_flat = z((tot + 3) // 4 * 4)
        self.noise: Dict[str, torch.Tensor] = {}
        o = 0
        for k in NOISE_KEYS:
            n = int(torch.Size(shapes[k]).numel())
            self.noise[k] = self.noise_flat[o:o + n].view(shapes[k])
            o += n

        z0 = int(torch.Size(shapes["eps_vae"]).numel())
        self._z_all = self.noise_flat[z0:z0 + sum(int(torch.Size(shapes[k]).numel()) for k in ("z_c", "z_cc", "z_actor"))]
        assert self._z_all.data_ptr() == self.noise["z_c"].data_ptr()
        twin = lambda mod: net_desc_seq(mod.all_nets(), 1.0)  # noqa: E731
        self.d_actor = net_desc_seq([m.actor.pi], 1.0)
        self.d_actor_old = net_desc_seq([m.actor_old.pi], 1.0)
        self.d_critic, self.d_cost = twin(m.critic), twin(m.cost_critic)
        self.d_critic_old, self.d_cost_old = twin(m.critic_old), twin(m.cost_critic_old)
        self.d_enc, self.d_dec = vae_enc_desc(m.vae), vae_dec_desc(m.vae)
        g = m.groups
        m.repack()

        # vae phase
        self.r_enc, self.r_dec = MlpRun(self.d_enc, B, True, dev), MlpRun(self.d_dec, B, True, dev)
        self.z, self.du, self.dhead_enc = z(B, Lz), z(1, B, ad), z(1, B, 2 * Lz)
        self.r_dec.setup_backward(self.du, dx_cols=(od, Lz))
        self.r_enc.setup_backward(self.dhead_enc)
        pl = self.plan = P.bcql_plan(od, ad, B, int(m.vae_hidden_sizes), N, seeds=G.SEEDS and G.VAE_TAILS and G.VAE_NS_AUTO)
        if pl.vae_dw_tile:
            # 400-wide layers = 5 x 5 column blocks: 80 x 80 tiles, 3 row splits per 2048 rows (engine/cpq.py)
            self.p_vae = DwPlan(g["vae"], self.r_enc.dw_entries() + self.r_dec.dw_entries(), B, dev,
                                n_splits=max(1, (3 * B) // 2048), tile_blocks=5)
        else:
            self.p_vae = DwPlan(g["vae"], self.r_enc.dw_entries() + self.r_dec.dw_entries(), B, dev)

        # target pipeline buffers (shared by the critic and the cost-critic phases)
        NB = N * B
        tr = pl.target_tile  # 80: mlp_fwd_nb_kernel (round 3: 564 vs 554.5 steps/s at C3); 0: tile kernel
        self.r_dec_t = MlpRun(self.d_dec, NB, False, dev, tile_rows=tr)
        self.r_actor_old_t = MlpRun(self.d_actor_old, NB, False, dev, tile_rows=tr)
        self.a_t = z(NB, ad)
        self.r_qold_t = MlpRun(self.d_critic_old, NB, False, dev, tile_rows=tr)
        self.r_qcold_t = MlpRun(self.d_cost_old, NB, False, dev, tile_rows=tr)
        # second set of pipeline buffers: the cost-critic phase runs on a side graph branch beside the critic phase
        self.r_dec_t2 = MlpRun(self.d_dec, NB, False, dev, tile_rows=tr)
        self.r_actor_old_t2 = MlpRun(self.d_actor_old, NB, False, dev, tile_rows=tr)
        self.a_t2 = z(NB, ad)

        dws = pl.dw_splits or None  # (engine/plan.py)
        self.r_critic = MlpRun(self.d_critic, B, True, dev)
        self.dq = z(2 * nq, B, 1)
        self.r_critic.setup_backward(self.dq)
        self.p_critic = DwPlan(g["critic"], self.r_critic.dw_entries(), B, dev, n_splits=dws)
        self.r_cost = MlpRun(self.d_cost, B, True, dev)
        self.dqc = z(2 * nqc, B, 1)
        self.r_cost.setup_backward(self.dqc)
        self.p_cost = DwPlan(g["cost_critic"], self.r_cost.dw_entries(), B, dev, n_splits=dws)

        # actor phase
        self.r_dec_b = MlpRun(self.d_dec, B, False, dev)
        self.r_actor = MlpRun(self.d_actor, B, True, dev)
        self.a_pi = z(B, ad)
        self.r_pi_q = MlpRun(concat_nets(self.d_critic, self.d_cost), B, True, dev)
        self.dq_pi = z(2 * nq + 2 * nqc, B, 1)
        self.r_pi_q.setup_backward(self.dq_pi, need_dz=False, dx_cols=(od, ad))
        self.dt = z(1, B, ad)
        self.pi_means = z(4)
        self.r_actor.setup_backward(self.dt)
        self.p_actor = DwPlan(g["actor"], self.r_actor.dw_entries(), B, dev, n_splits=dws)
        # batch-sum losses on a grid beyond 2048 rows (one workgroup walking 4096 rows x 10 samples x 4 nets alone: 44 us)
        big = B > 2048
        self.ws_vae, self.ws_c, self.ws_cc = (G.loss_ws(dev) if big else None for _ in range(3))
        # loss seeds (round 4, as in engine/cpq.py): the VAE decoder's and the two online critic ensembles' backward
        # launches compute the gradient they start from -- vae_loss and the two bcq_critic_loss launches leave the chains
        self.seeds = None
        if G.SEEDS and G.VAE_TAILS:
            self.seeds = {
                "vae": G.seed_vae(self.act, self.r_enc.y[0], B, ad, Lz, m.beta, self.rows_global, G.SeedStat(dev, 1, B),
                                  self.st.stat_ptr("loss/loss_vae")),
                "critic": G.seed_bcq_critic(self.r_qold_t.y, nq, nq, N, self.rew, self.done, B, m.gamma, m.lmbda,
                                            self.rows_global, G.SeedStat(dev, 2 * nq, B),
                                            self.st.stat_ptr("loss/critic_loss")),
                "cost": G.seed_bcq_critic(self.r_qcold_t.y, nqc, nqc, N, self.cost, None, B, m.gamma, m.lmbda,
                                          self.rows_global, G.SeedStat(dev, 2 * nqc, B),
                                          self.st.stat_ptr("loss/cost_critic_loss")),
            }
        # round 5: the VAE phase as all-CU layer launches where the library takes the shape (glue.VaeNs, engine/cpq.py)
        self.vae_ns = None
        if self.seeds is not None and pl.vae_ns:
            self.vae_ns = G.VaeNs.build(self.r_enc, self.r_dec, self.obs, self.act, self.noise["eps_vae"], self.z,
                                        m.latent_dim, m.beta, self.rows_global, self.st.stat_ptr("loss/loss_vae"))
        # every dW plan of this engine is built: the slab epochs they were built against are recorded NOW (not at the
        # first step), so an engine that is constructed directly, never stepped and then superseded is flagged stale
        from .core import slab_epochs
        self._slab_epochs = slab_epochs(self.model)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.replay = None

    def _optim(self, name: str, plan: DwPlan, tau: float) -> None:
        plan.launch()
        grp = self.model.groups[name]
        if self.dist is not None:
            self.dist.allreduce_group(grp)
        grp.adam_step(self.model._lrs[name], self.st.ptr, tau=tau)

    def _targets(self, zkey: str, r_q: MlpRun, second: bool = False) -> torch.Tensor:
        """bcql.py:138-142: Q_old(obs', actor_old(obs', vae.decode(obs'))) on the N*B repeated rows."""
        m, N, NB = self.model, self.model.sample_action_num, self.model.sample_action_num * self.B
        r_dec, r_act, a_t = (self.r_dec_t2, self.r_actor_old_t2, self.a_t2) if second else \
            (self.r_dec_t, self.r_actor_old_t, self.a_t)
        dec = r_dec.forward(self.nobs, self.noise[zkey], map0=L.MAP_DIV, div0=N)[0]
        t = r_act.forward(self.nobs, dec, map0=L.MAP_DIV, div0=N)[0]
        G.bcq_perturb(dec, t, NB, m.action_dim, m.phi, m.max_action, a_t)
        return r_q.forward(self.nobs, a_t, map0=L.MAP_DIV, div0=N)

    def _update(self, name: str, tau: float) -> None:
        grp = self.model.groups[name]
        if self.dist is not None:
            self.dist.allreduce_group(grp)
        grp.adam_step(self.model._lrs[name], self.st.ptr, tau=tau)

    def head(self, device_noise: bool) -> None:
        """The part of a step that depends on nothing the PREVIOUS step's critic / actor phases write: prologue (tick +
        minibatch gather + noise), the latent clamp, and the whole VAE phase (``vae_loss`` bcql.py:122-132 + its optimizer
        step).  Its inputs are this engine's own batch / noise buffers and the VAE, whose last reader in a step is the
        second target pipeline's decoder launch -- so in a pipelined graph (engine/pipeline.py) the NEXT step's head runs
        on the side queue under this step's actor phase."""
        m, st, nz, B = self.model, self.st, self.noise, self.B
        od, ad, Lz, N = m.state_dim, m.action_dim, m.latent_dim, m.sample_action_num
        rg = self.rows_global
        st.prologue(self.replay, (self.obs, self.nobs, self.act, self.rew, self.cost, self.done), self.noise_flat, self.seed, device_noise)
        # net.py:334-335 clamps the latent draws: z_c | z_cc | z_actor are adjacent in the flat noise buffer -- ONE launch
        # over the range instead of three (round 5: 3 x 5.2 us on the step's head, profiles/r4_bcql_timeline.txt)
        G.clamp_(self._z_all, -0.5, 0.5)

        sd = self.seeds
        if self.vae_ns is not None:
            self.vae_ns.forward()
            self.vae_ns.backward()
        else:
            head = G.vae_encode(self.r_enc, self.obs, self.act, nz["eps_vae"], Lz, self.z)
            u = self.r_dec.forward(self.obs, self.z)[0]
        if self.vae_ns is not None:
            pass
        elif sd is not None:
            self.r_dec.backward_dz(tail=G.vae_latent_bwd_tail(head, nz["eps_vae"], Lz, m.beta, rg, self.dhead_enc),
                                   seed=sd["vae"])
        else:
            G.vae_loss(u, self.act, head, B, ad, Lz, m.beta, rg, self.du, st.stat_ptr("loss/loss_vae"), ws=self.ws_vae)
            G.vae_decoder_backward(self.r_dec, head, nz["eps_vae"], Lz, m.beta, rg, self.dhead_enc)
        if self.vae_ns is None:
            self.r_enc.backward_dz()
        self._optim("vae", self.p_vae, 0.0)

    def body(self, device_noise: bool, par: Optional[Branches] = None, nxt: Optional["BCQLEngine"] = None,
             head_done: bool = False) -> None:
        """``par`` (graph capture): cost_critic_loss (bcql.py:157-179) reads only the updated VAE, actor_old and
        cost_critic_old -- nothing the critic phase writes -- so it runs on a side branch beside critic_loss; its
        optimizer step waits for the join (it Polyak-updates nothing the critic branch reads, but a data-parallel
        all-reduce must stay on the capture stream).

        ``nxt`` / ``head_done`` (engine/pipeline.py, several steps per graph): behind the join the side queue is idle for
        the whole actor phase (profiles/r5_timeline_c3.txt: 1318-1540 us), and the main queue runs the next step's VAE phase
        alone (0-205 us) -- so the NEXT step's ``head()``, on the twin engine ``nxt``'s buffers and step state, is issued
        on the side branch right behind the join; the next step then runs with ``head_done=True``."""
        par = par or Branches(False)
        m, st, nz, B = self.model, self.st, self.noise, self.B
        od, ad, Lz, N = m.state_dim, m.action_dim, m.latent_dim, m.sample_action_num
        nq, nqc, rg = m.num_q, m.num_qc, self.rows_global
        sd = self.seeds
        assert nxt is None or self.dist is None, "pipelined steps are a single-GPU plan"
        if not head_done:
            self.head(device_noise)
        # round 3 (profiles/r3_bcql_timeline.txt): both branches are linear chains from here (a side branch forked BEFORE
        # the VAE phase, to run the online forwards beside it, made the graph executor put both 950 us target pipelines
        # on one queue: 1881 vs 1788 us).  The online critics' forwards (critic / cost critic on (obs, act),
        # bcql.py:144,167: no dependency on anything the step computes) go out as ONE paired launch at the head of the
        # side branch instead of two launches behind the pipelines; the head of actor_loss (bcql.py:183-187: needs the
        # updated VAE and the not-yet-updated actor only) runs at the head of THIS branch instead of at the side
        # branch's tail, which was the later one at the join.
        # (creation order matters to the executor: THIS branch's first launches are created before the side branch's,
        # so that it stays on the queue the VAE phase ran on and the side branch gets the second one)
        par.fork(0)
        dec = self.r_dec_b.forward(self.obs, nz["z_actor"])[0]
        t = self.r_actor.forward(self.obs, dec)[0]
        G.bcq_perturb(dec, t, B, ad, m.phi, m.max_action, self.a_pi)
        q_t = self._targets("z_c", self.r_qold_t)

        with par.on(0):
            q, qc = self.r_critic.forward_with((self.obs, self.act), self.r_cost, (self.obs, self.act))
            ev_on = par.mark(0)
            qc_t = self._targets("z_cc", self.r_qcold_t, second=True)
            if sd is not None:
                self.r_cost.backward_dz(seed=sd["cost"])
            else:
                G.bcq_critic_loss(qc_t, nqc, nqc, N, qc, 2 * nqc, self.cost, None, B, m.gamma, m.lmbda, rg, self.dqc,
                                  st.stat_ptr("loss/cost_critic_loss"), ws=self.ws_cc)
                self.r_cost.backward_dz()
            self.p_cost.launch()

        par.wait(ev_on)
        if sd is not None:
            self.r_critic.backward_dz(seed=sd["critic"])
        else:
            G.bcq_critic_loss(q_t, nq, nq, N, q, 2 * nq, self.rew, self.done, B, m.gamma, m.lmbda, rg, self.dq,
                              st.stat_ptr("loss/critic_loss"), ws=self.ws_c)
            self.r_critic.backward_dz()
        if self.dist is None:
            self._optim("critic", self.p_critic, m.tau)
        else:  # reduced together with the cost critic's gradient after the join: one collective instead of two
            self.p_critic.launch()
        par.join(0)
        if nxt is not None:  # (pipelined: the next step's prologue + VAE phase under this step's actor phase)
            with par.on(0):
                nxt.head(device_noise)
        if self.dist is None:
            self._update("cost_critic", m.tau)
        else:
            gc, gcc = m.groups["critic"], m.groups["cost_critic"]
            self.dist.all_reduce_many_([self.dist.reduce_local(gc), self.dist.reduce_local(gcc)])
            gc.adam_step(m._lrs["critic"], st.ptr, tau=m.tau)
            gcc.adam_step(m._lrs["cost_critic"], st.ptr, tau=m.tau)

        y = self.r_pi_q.forward(self.obs, self.a_pi)
        means, share = None, 1.0
        if self.dist is not None:  # the PID controller acts on the GLOBAL mean of qc_pi
            G.bcq_actor_sums(y[:2 * nq], nq, nq, y[2 * nq:], nqc, nqc, B, rg, self.pi_means)
            self.dist.all_reduce_(self.pi_means)
            means, share = self.pi_means, 1.0 / self.dist.world
        G.bcq_actor_loss(y[:2 * nq], nq, nq, y[2 * nq:], nqc, nqc, B, m.qc_thres, m.KP, m.KI, m.KD, rg, m.pid_state,
                         self.dq_pi[:2 * nq], self.dq_pi[2 * nq:], st.stat_ptr("loss/actor_loss"), means, share)
        self.r_pi_q.backward_dz()
        G.bcq_perturb_bwd(dec, t, self.r_pi_q.dx, 2 * nq + 2 * nqc, B, ad, m.phi, m.max_action, self.dt)
        self.r_actor.backward_dz()
        if self.dist is None:
            self._optim("actor", self.p_actor, m.tau)
        else:  # actor gradient and the per-rank partial statistics in one collective
            self.p_actor.launch()
            ga = m.groups["actor"]
            self.dist.all_reduce_many_([self.dist.reduce_local(ga), st.stats])
            ga.adam_step(m._lrs["actor"], st.ptr, tau=m.tau)
        if nxt is not None:
            par.join(0)  # the next step's critic / actor phases need its head (the VAE's optimizer step)

    def load_batch(self, observations, next_observations, actions, rewards, costs, done) -> None:
        load_into(((self.obs, observations), (self.nobs, next_observations), (self.act, actions),
                   (self.rew, rewards), (self.cost, costs), (self.done, done)))

    def _snapshot(self):
        m = self.model
        snap = {"pid": m.pid_state.clone(), "state": self.st.state.clone(), "host": self.st.host_step,
                "stats": self.st.stats.clone(), "ring": self.st.ring.clone()}
        for n, g in m.groups.items():
            snap[n] = (g.p.clone(), g.m.clone(), g.v.clone(), None if g.tgt is None else g.tgt.clone())
        return snap

    def _restore(self, snap) -> None:
        m = self.model
        m.pid_state.copy_(snap["pid"])
        self.st.state.copy_(snap["state"]); self.st.stats.copy_(snap["stats"]); self.st.ring.copy_(snap["ring"])
        self.st.host_step = snap["host"]
        for n, g in m.groups.items():
            p, mm, v, t = snap[n]
            g.p.copy_(p); g.m.copy_(mm); g.v.copy_(v)
            if t is not None:
                g.tgt.copy_(t)
        m.repack()

    def capture(self) -> None:
        """(A few captures, the fastest graph kept: core.pick_fastest.)"""
        from .core import CAPTURE_TRIES, pick_fastest

        def once():
            snap = self._snapshot()
            par = Branches(True, 1)
            g, arena = capture_step(self.st.state.device, lambda: self.body(True), lambda: self.body(True, par))
            torch.cuda.synchronize()
            self._restore(snap)
            return g, par, arena  # (the side stream and the argument blocks stay alive with the graph)

        (self.graph, self._par, self._arena), self.capture_ms = pick_fastest(once, lambda c: c[0].replay(), self._snapshot,
                                                                              self._restore, CAPTURE_TRIES)

    def attach_replay(self, store) -> None:
        """Sample minibatches on device from ``store`` (common/replay.py) inside the step itself."""
        self.replay = store
        self.graph = None
        self._pipe = None

    def steps_replay(self, n: int, steps_per_graph: Optional[int] = None) -> None:
        """EXACTLY ``n`` train steps on minibatches drawn on device from the attached replay store.  Where the plan says so
        (``plan.steps_per_graph`` > 1, single GPU) whole multiples go through graphs of that many steps, software-pipelined
        across steps (engine/pipeline.py: bit-equal to ``n`` calls of ``step_replay()``); the remainder through the
        one-step graph.  The loop of examples/train/train_cpq.py:138-144 / train_bcql.py:142-148 with the DataLoader
        folded into the step."""
        spg = int(self.plan.steps_per_graph if steps_per_graph is None else steps_per_graph)
        if spg <= 1 or self.dist is not None:
            for _ in range(int(n)):
                self.step_replay(True)
            return
        pipe = getattr(self, "_pipe", None)
        if pipe is None or pipe.n != spg:
            from .pipeline import PipelinedSteps
            pipe = self._pipe = PipelinedSteps(self, spg)
        pipe.run(n)

    def step_replay(self, use_graph: bool = True) -> None:
        """One train step on a minibatch drawn on device from the attached replay store."""
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
