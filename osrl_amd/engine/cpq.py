"""The CPQ train step as a static launch plan on MI355X.

Follows ``CPQTrainer.train_one_step`` (osrl/algorithms/cpq.py:294-313) phase by phase:
``vae_loss`` :125-135 -> ``critic_loss`` :137-153 -> ``cost_critic_loss`` :155-201 ->
``actor_loss`` :203-222 -> ``sync_weight`` :224-230 (fused into each group's Adam kernel).

Every buffer is allocated once; one step is a fixed sequence of ~40 kernel launches with no host
synchronisation, so it is captured in a hipGraph (``torch.cuda.CUDAGraph``) and replayed.
Common sub-expressions the reference recomputes are evaluated once (exact, not approximate):
  * the actor trunk on ``next_observations`` (cpq.py:141 and :159 differ only by the noise draw);
  * the actor trunk on ``observations`` (cpq.py:164 and :209: the actor is not updated in between).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from .. import _lib as L
from ..common.net import actor_head_desc, net_desc_seq, vae_dec_desc, vae_enc_desc
from . import glue as G
from . import plan as P
from .core import ArgArena, Branches, DwPlan, MlpRun, StepState, concat_nets, load_into, check_plans_current, graph_capture

# Plan choices that depend on the shape live in engine/plan.py (cpq_plan: head tails, dW tiles, the all-CU VAE launches;
# the measurements behind each rule are in DESIGN_LOG.md).  What is left here are lab switches of the plan's STRUCTURE:
# Data parallel, round 5: the two collectives that feed only the side branch (the VAE gradient's all-reduce, the KL values'
# all-gather) issued from a branch of their own / the side branch instead of the main one.  Order-safe without a second
# communicator: every rank ISSUES the four collectives in the same program order and RCCL executes them on its own stream
# in that order; only the streams that wait for them differ.  Built, verified (world-2 / world-8 in-process capture tests)
# and measured with forced data parallelism on one rank: SLOWER, C2 1991 vs 2021 steps/s, C4 2065 vs 2213
# (gpurun_out/r5d): the extra branch and the collectives' fork / join edges on the side queue cost the graph executor more
# than the two 17 us collectives they take off the main chain.  Default "0" = round 4's placement (all four on the main
# branch).
DP_SIDE_COLL = P.knob("OSRL_DP_SIDE_COLL", "0", "DP: VAE all-reduce / KL gather issued off the main branch") == "1"
# (pipelined graphs: where the next prologue sits and whether the steps of a graph are joined are plan fields --
# engine/plan.py pipe_prologue / pipe_no_join; the lab switch OSRL_PIPE_DUAL=side keeps round 6's side-branch dual step)
PIPE_DUAL_SIDE = P.knob("OSRL_PIPE_DUAL", "auto") == "side"
# (no-join graphs with plan.vae_adam_side: the edge from the VAE's optimizer step on step k's side branch to the main chain,
# whose next VAE phase reads what it wrote -- in front of the actor group's Adam, the last launch of step k's main chain
# (actor) / at the head of step k+1's main chain (next) / none (0: lab only -- the second session's graphs, ordered by ~100 us
# of timing slack and nothing else))
# (no-join graphs with the VAE's optimizer step on the MAIN chain -- C4: step k's N*B-row encoder launch, late on its side branch,
# reads the weights step k+1's optimizer step rewrites ~280 us later, and the main chain's first edge from that branch is the
# wait for the action draws BEHIND that optimizer step.  1: an event behind the encoder launch, waited for in front of the
# optimizer step; 0 (lab): unordered, as the second session's graphs were)
VAE_WAR_EDGE = P.knob("OSRL_VAE_WAR_EDGE", "actor", "no-join graphs, VAE Adam on the main chain: ordered behind the previous step's "
                      "N*B-row encoder launch by an edge to that step's actor Adam (actor) / to the VAE Adam itself (vae) / by "
                      "timing slack (0, lab)")
VAE_ADAM_EDGE = P.knob("OSRL_VAE_ADAM_EDGE", "actor", "no-join graphs, VAE Adam on the side branch: its edge to the main chain "
                       "in front of the actor group's Adam (actor) / at the head of the next step (next) / none (0, lab)")
# (lab, no-join graphs: the actor group's dW + Adam of step k at the head of step k+1's side branch instead of the tail of
# step k's main chain -- nothing on the main chain reads the actor before the next trunk launch, which is on that branch)
PIPE_ACTOR_SIDE = P.knob("OSRL_PIPE_ACTOR", "main", "no-join pipelined steps: the actor group's dW + Adam on the main chain (main) / "
                         "carried to the head of the next step's side branch (side)") == "side"
STAT_KEYS = ["loss/loss_vae", "loss/critic_loss", "loss/cost_critic_loss", "loss/alpha_value", "loss/actor_loss"]
NOISE_KEYS = ["eps_vae", "eps_next_c", "eps_next_cc", "eps_ood", "eps_actor"]


class CPQEngine:
    def __init__(self, model, batch_size: int, rows_global: int = 0, seed: int = 0, dist=None):
        m = self.model = model
        B = self.B = int(batch_size)
        self.rows_global = int(rows_global)
        # data parallel: the ranks' rows are different samples of one global batch -> independent noise streams
        self.seed = seed if dist is None else dist.rank_seed(seed)
        self.dist = dist
        dev = torch.device(m.device)
        self.dev = dev
        od, ad, Lz, N = m.state_dim, m.action_dim, m.latent_dim, m.sample_action_num
        f = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f)  # noqa: E731
        self.st = StepState(dev, STAT_KEYS)
        self._prologue_covered = False
        self._actor_pending = False
        self._polyak_pending = False
        self._ev_prologue = None    # (pipelined graphs, plan.pipe_no_join: event behind the next step's prologue)
        # (tests/test_gpu_pipeline.py test_unjoined_graphs_are_ordered_by_edges_not_by_timing: a spin kernel of this many
        # cycles on the side branch in front of its second half -- VAE Adam / N*B-row encoder launch -- of every step)
        self._stress_spin = int(P.knob("OSRL_STRESS_SPIN_CYCLES", "0", "tests: spin kernel (cycles) in front of the side branch's second half"))
        self._stress_at = P.knob("OSRL_STRESS_SPIN_AT", "second", "tests: ... (second) / at the head of the side branch (head) / at the "
                                 "head of the main chain (main)")
        self._ev_vae_adam = None    # (... and behind this step's VAE Adam where that runs on the side branch)
        self._ev_enc_ood = None     # (... and behind this step's N*B-row encoder launch where the VAE Adam runs on the main chain)
        self._dual_pending = False  # (this step's dual step is still to be issued by the next step of the graph)
        nq, nqc = m.num_q, m.num_qc
        c_hidden = [int(l.out_features) for l in m.cost_critic_old.q_nets[0] if isinstance(l, torch.nn.Linear)][:-1]
        pl = self.plan = P.cpq_plan(od, ad, B, int(m.vae_hidden_sizes), N, seeds=G.SEEDS and G.VAE_TAILS and max(nq, nqc) <= 4
                                    and G.VAE_NS_AUTO, c_hidden=c_hidden)
        # the OOD rows as a set (plan.ood_rows) is a single-GPU plan: the data-parallel step keeps the masked mean over all rows
        self.ood_rows = bool(pl.ood_rows) and dist is None

        # static inputs (a replayed graph reads these addresses)
        self.obs, self.nobs, self.act = z(B, od), z(B, od), z(B, ad)
        self.rew, self.cost, self.done = z(B), z(B), z(B)
        # one flat noise buffer -> one Philox launch per step
        shapes = {"eps_vae": (B, Lz), "eps_next_c": (B, ad), "eps_next_cc": (B, ad), "eps_ood": (N, B, ad),
                  "eps_actor": (B, ad)}
        tot = sum(int(torch.Size(s).numel()) for s in shapes.values())
        self.noise_flat = z((tot + 3) // 4 * 4)
        self.noise: Dict[str, torch.Tensor] = {}
        o = 0
        for k in NOISE_KEYS:
            n = int(torch.Size(shapes[k]).numel())
            self.noise[k] = self.noise_flat[o:o + n].view(shapes[k])
            o += n

        # network descriptors (pointers into the flat groups)
        self.d_actor = actor_head_desc(m.actor)
        self.d_critic = net_desc_seq(list(m.critic.q_nets), 1.0)
        self.d_cost = net_desc_seq(list(m.cost_critic.q_nets), 1.0)
        self.d_critic_old = net_desc_seq(list(m.critic_old.q_nets), 1.0)
        self.d_cost_old = net_desc_seq(list(m.cost_critic_old.q_nets), 1.0)
        self.d_enc = vae_enc_desc(m.vae)
        self.d_dec = vae_dec_desc(m.vae)
        g = m.groups
        m.repack()

        # ---- vae phase
        self.r_enc = MlpRun(self.d_enc, B, True, dev)
        self.r_dec = MlpRun(self.d_dec, B, True, dev)
        self.z = z(B, Lz)
        self.du = z(1, B, ad)
        self.dhead_enc = z(1, B, 2 * Lz)
        self.r_dec.setup_backward(self.du, dx_cols=(od, Lz))
        self.r_enc.setup_backward(self.dhead_enc)
        # 1024 rows per split-K slab (the default policy gives 256): 140 tiles x 2 splits = 280 workgroups fit the 512
        # resident slots in one round (x 4 splits = 560: a second, almost empty round), and the Adam kernel sums two
        # slabs instead of four.  In the step: 2175 steps/s vs 2135 (4 splits), 2140 (3), 2015 (1); the other groups
        # measure best at the default 256 rows (2175 vs 2110 at 512, 2010 at 1024)
        # round 3: 400-wide layers are 25 = 5 x 5 column blocks -- on 80 x 80 tiles with a flat (tile, split) work list
        # the VAE's dW is 250 even workgroups in one round (core.DwPlan tile_blocks; OSRL_VAE_DW_T5=0: the 64 x 64 form)
        if pl.vae_dw_tile:
            self.p_vae = DwPlan(g["vae"], self.r_enc.dw_entries() + self.r_dec.dw_entries(), B, dev,
                                n_splits=pl.vae_dw_splits, tile_blocks=pl.vae_dw_tile)  # 70 tiles x 3 splits = 210 workgroups: one round (tools/dw_bench.py)
        else:
            self.p_vae = DwPlan(g["vae"], self.r_enc.dw_entries() + self.r_dec.dw_entries(), B, dev,
                                n_splits=pl.vae_dw_splits)

        # ---- critic phase
        self.r_actor_next = MlpRun(self.d_actor, B, False, dev)
        self.a_next = z(B, ad)
        self.r_old_next = MlpRun(concat_nets(self.d_critic_old, self.d_cost_old), B, False, dev)
        self.r_critic = MlpRun(self.d_critic, B, True, dev)
        self.dq = z(nq, B, 1)
        self.r_critic.setup_backward(self.dq)
        # round 4: the two critic groups' dW on 32 x 32 tiles x 2 row splits (17 KB of LDS, 154 registers per lane): such
        # workgroups fit on a CU beside the 8-wave N*B-row encoder launch (130 KB, 2 x 152 registers per SIMD), which a
        # 64 x 64 tile's 68 KB / 304 registers do not -- the cost critics' dW runs beside that launch on the main branch
        # (84 -> 65 us there; C2 2245 -> 2260-2267 steps/s with both groups, gpurun_out/r4c)
        t_def, s_def = ("2", 2) if pl.small_dw else ("0", None)
        self.p_critic = DwPlan(g["critic"], self.r_critic.dw_entries(), B, dev,
                               tile_blocks=int(P.knob("OSRL_DW_T_CRITIC", t_def, "dW tile of the critic group (16-blocks)")),
                               n_splits=(int(P.knob("OSRL_DW_S_CRITIC", "0")) if P.knob_set("OSRL_DW_S_CRITIC") else s_def))

        # ---- cost-critic phase
        self.a_next2 = z(B, ad)
        self.r_costold_next = MlpRun(self.d_cost_old, B, False, dev)
        self.r_actor_obs = MlpRun(self.d_actor, B, True, dev)  # also the actor forward of the actor phase
        self.sampled = z(N * B, ad)
        # Both N*B-row forwards (target cost critics beside the VAE phase, VAE encoder beside the cost-critic phase)
        # take the 80-row one-workgroup-per-CU kernel (csrc/mlp.hip mlp_fwd_nb_kernel): 73 / 71 us alone vs 98 / 88 us
        # for 32-row tiles.  Its 84 KB (256-wide) / 134 KB (400-wide) of LDS per workgroup leave the chain's launches
        # room on every CU.  (Until the kernel's epilogue / bias / stage-in rework of round 2 the capped 32-row tile
        # loop was the better neighbour for the VAE phase: 1968 vs 1876 steps/s; now 80-row tiles give 2095 vs 2020.
        # OSRL_OOD_TILE=0 OSRL_OOD_WG_CAP=512 restores the old form.)
        ood_tile = pl.ood_tile
        self.r_costold_ood = MlpRun(self.d_cost_old, N * B, False, dev,
                                    wg_cap=int(P.knob("OSRL_OOD_WG_CAP", "0", "workgroup cap of the N*B cost-critic launch")),
                                    tile_rows=ood_tile)
        self.r_enc_ood = MlpRun(self.d_enc, N * B, False, dev,
                                wg_cap=int(P.knob("OSRL_ENC_WG_CAP", "0", "workgroup cap of the N*B encoder launch")),
                                tile_rows=int(P.knob("OSRL_ENC_TILE", str(ood_tile or 80), "row tile of the N*B encoder launch")))
        # shared-observation tiles of the two N*B-row launches (plan.ood_share): their rows are the B observations N times over
        # (cpq.py:164-176), so the observation part of layer 0 runs once per observation of a tile (osrl_rows_t.share0)
        self.pre_cost = self.pre_enc = 0
        if pl.ood_share and ood_tile == 80:
            if not self.ood_rows:  # (a row SET has no tiles of shared observations)
                self.pre_cost = self.r_costold_ood.share_k16(od, B, N)
            if not self.ood_rows or P.knob("OSRL_OOD_ROWS_ENC_SHARE", "1", "plan.ood_rows: the N*B-row encoder launch keeps its shared-observation tiles: 1 / 0") == "1":
                self.pre_enc = self.r_enc_ood.share_k16(od, B, N)
        self.kl = z(N * B)
        self.quant = z(4)
        self.ood_mean = z(4)
        self.ood_list = torch.zeros(N * B, dtype=torch.int32, device=dev)  # plan.ood_rows: the rows with KL >= quantile, ascending
        self.ood_count = torch.zeros(4, dtype=torch.int32, device=dev)
        self.r_cost = MlpRun(self.d_cost, B, True, dev)
        self.dqc = z(nqc, B, 1)
        self.r_cost.setup_backward(self.dqc)
        self.p_cost = DwPlan(g["cost_critic"], self.r_cost.dw_entries(), B, dev,
                             tile_blocks=int(P.knob("OSRL_DW_T_COST", t_def, "dW tile of the cost-critic group (16-blocks)")),
                             n_splits=(int(P.knob("OSRL_DW_S_COST", "0")) if P.knob_set("OSRL_DW_S_COST") else s_def))

        # ---- actor phase
        self.a_pi, self.tanh_u = z(B, ad), z(B, ad)
        self.r_pi_q = MlpRun(concat_nets(self.d_critic, self.d_cost), B, True, dev, save_nets=list(range(nq)))
        self.dq_pi = z(nq, B, 1)
        self.r_pi_q.setup_backward(self.dq_pi, need_dz=False, dx_cols=(od, ad))
        self.dhead_actor = z(1, B, 2 * ad)
        self.r_actor_obs.setup_backward(self.dhead_actor)
        self.p_actor = DwPlan(g["actor"], self.r_actor_obs.dw_entries(), B, dev)

        # ---- loss seeds (round 4): each backward launch computes the gradient it starts from (glue.seed_*,
        # osrl_mlp_backward_dz_seed) -- the five loss launches between a forward and a backward leave the chains
        rg = self.rows_global
        st = self.st
        self.seeds = None
        if G.SEEDS and G.VAE_TAILS and max(nq, nqc) <= 4:
            y_old, y_pi = self.r_old_next.y, self.r_pi_q.y
            self.seeds = {
                "vae": G.seed_vae(self.act, self.r_enc.y[0], B, ad, Lz, m.beta, rg, G.SeedStat(dev, 1, B),
                                  st.stat_ptr("loss/loss_vae")),
                "critic": G.seed_cpq_critic(y_old[:nq], nq, y_old[nq:], nqc, self.rew, self.done, B, m.gamma, m.q_thres,
                                            rg, G.SeedStat(dev, nq, B), st.stat_ptr("loss/critic_loss")),
                "cost": G.seed_cpq_cost(self.r_costold_next.y, nqc, self.cost, B, m.gamma, rg, G.SeedStat(dev, nqc, B),
                                        st.stat_ptr("loss/cost_critic_loss")),
                "actor": G.seed_cpq_actor(y_pi[:nq], nq, y_pi[nq:], nqc, B, m.q_thres, rg, G.SeedStat(dev, nq, B),
                                          st.stat_ptr("loss/actor_loss")),
                "head": G.seed_gauss_head(self.noise["eps_actor"], self.tanh_u, self.r_pi_q.dx, nq, B, m.max_action),
            }

        # round 5: the VAE phase's forward / backward as all-CU layer launches (csrc/vae_ns.hip, glue.VaeNs) where the plan
        # says so (engine/plan.py vae_ns_auto: by measurement) and the library takes the shape
        self.vae_ns = None
        if self.seeds is not None and pl.vae_ns:
            self.vae_ns = G.VaeNs.build(self.r_enc, self.r_dec, self.obs, self.act, self.noise["eps_vae"], self.z, Lz,
                                        m.beta, rg, st.stat_ptr("loss/loss_vae"))

        # every dW plan of this engine is built: the slab epochs they were built against are recorded NOW (not at the
        # first step), so an engine that is constructed directly, never stepped and then superseded is flagged stale
        from .core import slab_epochs
        self._slab_epochs = slab_epochs(self.model)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.replay = None
        self.parallel_branches = True
        self._graph_failed = False
        self._probe = None

    # ------------------------------------------------------------------ #
    def _update(self, name: str, tau: float) -> None:
        """(data parallel: all-reduce of the flat gradient, then) the fused Adam + Polyak + repack of one group."""
        m = self.model
        grp = m.groups[name]
        if self.dist is not None:
            self.dist.allreduce_group(grp)
        grp.adam_step(m._lrs[name], self.st.ptr, tau=tau)

    def _optim(self, name: str, plan: DwPlan, tau: float) -> None:
        if name == "vae":
            self._pr("vae_dw", 0)
        if self.dist is None and plan.can_fuse_adam():  # dW and the optimizer step in one launch (same bits)
            plan.launch_adam(self.model._lrs[name], self.st.ptr, tau=tau)
            if name == "vae":
                self._pr("vae_dw", 1)
            return
        plan.launch()
        if name == "vae":
            self._pr("vae_dw", 1)
        self._update(name, tau)

    def _pr(self, site: str, i: int) -> None:
        """bench.py's in-step probe: HIP events (on the launching stream) around a named launch of the step body."""
        if self._probe is not None and site in self._probe:
            p = self._probe[site]
            if isinstance(p, int):  # device address of three uint64: stamps INSIDE the captured graph (osrl_stamp_realtime)
                cs = torch.cuda.current_stream().cuda_stream
                L.check(L.load().osrl_stamp_realtime(p + 8 * i, cs), "osrl_stamp_realtime")
                if i == 1:  # a second stamp right behind the closing one: the cost of a stamp itself (bench.py subtracts it)
                    L.check(L.load().osrl_stamp_realtime(p + 16, cs), "osrl_stamp_realtime")
            else:
                p[i].record()

    def prologue(self, device_noise: bool) -> None:
        """tick + minibatch gather + Philox noise of ONE step into this engine's buffers (one launch)."""
        self.st.prologue(self.replay, (self.obs, self.nobs, self.act, self.rew, self.cost, self.done), self.noise_flat,
                         self.seed, device_noise)

    def body(self, device_noise: bool, par: Optional[Branches] = None, nxt: Optional["CPQEngine"] = None,
             prologue_done: bool = False, prev: Optional["CPQEngine"] = None) -> None:
        """One step, single GPU or data parallel (``self.dist``): the launch plan below.

        ``nxt`` / ``prologue_done`` (engine/pipeline.py, several steps per graph): the NEXT step's prologue -- into the
        twin engine ``nxt``'s buffers and step state -- is issued at the tail of THIS step's side branch, between the N*B-row
        encoder launch and the single-workgroup OOD statistic (C2: +1.2 .. +2.6 % against +0.5 % behind it, gpurun_out/r6e);
        the next step is then run with ``prologue_done=True``.  (Measured and removed, round 6: the two streams SWAPPING roles from step to step, so that
        no cross-queue edge stands in front of the next VAE phase -- the graph executor then opens a third queue for the
        swapped main chain, C2 1884-2062 vs 2300-2325 steps/s, and a replay of that graph segfaulted in the runtime:
        profiles/r6_timeline_4step_swap_c2.txt, DESIGN_LOG round 6.)

        What the plan exploits (cpq.py:155-201): everything the OOD penalty is made of -- the N*B sampled actions, the
        target cost critics on them, the VAE encoder on them, the KL rows, their 0.75-quantile, ``qc_ood`` -- sits
        under ``torch.no_grad()`` and enters the cost-critic loss as ``- exp(log_alpha) * (qc_ood.mean() - thres)``,
        a term with NO gradient to any network.  It moves ``log_alpha`` and the logged loss, nothing else.  So the
        69 % of the step's FLOPs that live in the two N*B-row launches are not on the path of any parameter update,
        and the plan keeps them off the latency chain:

          main  : vae phase -> cost-critic phase (MSE part only) -> [critic updated] -> actor phase
          side  : actor forwards + heads -> target cost critics on the N*B rows -> critic phase ->
                  (vae updated) encoder on the N*B rows -> KL -> quantile -> qc_ood mean -> dual step + logged loss

        The side branch joins at the END of the step.  Ordering constraints kept by events: the cost-critic Adam also
        Polyak-updates ``cost_critic_old``, so it waits for the side branch's last reader of the targets; the encoder on
        the N*B rows waits for the VAE's Adam; the actor phase waits for the critic's Adam."""
        dp = self.dist
        m, st, nz, B = self.model, self.st, self.noise, self.B
        od, ad, Lz, N = m.state_dim, m.action_dim, m.latent_dim, m.sample_action_num
        nq, nqc, rg = m.num_q, m.num_qc, self.rows_global
        par = par or Branches(False)
        assert nxt is None or dp is None, "pipelined steps are a single-GPU plan"
        if not prologue_done:
            self.prologue(device_noise)
        # plan.pipe_no_join (pipelined graphs): the steps of a graph are not joined.  The main chain of step k+1 waits for
        # its prologue only (issued on step k's side branch: ``_ev_prologue``), and step k's dual step runs at the head of
        # step k+1's side branch, right behind the fork -- the one place of the side branch that already has an edge from
        # the END of step k's main chain (so both halves of the logged cost loss are there) without a new mid-chain edge.
        no_join = self.plan.pipe_no_join and par.enabled and dp is None
        # (read per capture, not per process: the ordering test builds graphs with and without these edges)
        VAE_ADAM_EDGE, VAE_WAR_EDGE = P.knob("OSRL_VAE_ADAM_EDGE", "actor"), P.knob("OSRL_VAE_WAR_EDGE", "actor")
        carried = prev if (no_join and prev is not None and prev._dual_pending) else None
        self._ev_prologue, self._prologue_covered = None, False
        if carried is not None and carried._ev_prologue is not None:
            par.wait(carried._ev_prologue)
        if carried is not None and carried._ev_vae_adam is not None and VAE_ADAM_EDGE == "next":
            par.wait(carried._ev_vae_adam)  # (this step's VAE phase reads what that optimizer step wrote)
        self._ev_vae_adam = None
        par.fork(0)
        if self._stress_spin and par.enabled and self._stress_at == "main":
            torch.cuda._sleep(self._stress_spin)  # (tests: the main chain arrives late)
        # ---- main: vae_loss  (cpq.py:125-135)
        sd = self.seeds
        if self.vae_ns is not None:  # five all-CU layer launches instead of the four fused ones (same buffers)
            self.vae_ns.forward()
            self.vae_ns.backward()
            head = self.r_enc.y[0]
        else:
            head = G.vae_encode(self.r_enc, self.obs, self.act, nz["eps_vae"], Lz, self.z)
            u = self.r_dec.forward(self.obs, self.z)[0]
        if self.vae_ns is not None:
            pass
        elif sd is not None:  # reconstruction gradient + the logged loss by the decoder's backward launch itself
            self.r_dec.backward_dz(tail=G.vae_latent_bwd_tail(head, nz["eps_vae"], Lz, m.beta, rg, self.dhead_enc),
                                   seed=sd["vae"])
        else:
            G.vae_loss(u, self.act, head, B, ad, Lz, m.beta, rg, self.du, st.stat_ptr("loss/loss_vae"))
            G.vae_decoder_backward(self.r_dec, head, nz["eps_vae"], Lz, m.beta, rg, self.dhead_enc)
        if self.vae_ns is None:
            self.r_enc.backward_dz()
        if dp is not None and par.enabled and len(par.side) > 1 and DP_SIDE_COLL:
            # data parallel, round 5: nothing on the main branch reads the VAE's parameters in this step (its only reader
            # is the N*B-row encoder launch of the side branch), so the gradient's all-reduce and the optimizer step
            # leave the critical chain: a short branch of their own behind the dW launch.  The collective is still ISSUED
            # here, first of the step's four, on every rank (RCCL runs them on its own stream in issue order)
            self._pr("vae_dw", 0)
            self.p_vae.launch()
            self._pr("vae_dw", 1)
            par.fork(1)
            with par.on(1):
                self._update("vae", 0.0)
                ev_vae = par.mark(1)
        else:
            # (single GPU, plan.vae_adam_side: the VAE's optimizer step at the head of the side branch's second half, in
            # front of its only reader, instead of on the main chain -- +0.7 % at C2, DESIGN_LOG round 5)
            vae_adam_side = dp is None and par.enabled and self.plan.vae_adam_side and not self.p_vae.can_fuse_adam()
            if vae_adam_side:
                self._pr("vae_dw", 0)
                self.p_vae.launch()
                self._pr("vae_dw", 1)
            else:
                if carried is not None and carried._ev_enc_ood is not None:  # (OSRL_VAE_WAR_EDGE=vae)
                    par.wait(carried._ev_enc_ood)  # (the previous step's last reader of the weights this step rewrites)
                    carried._ev_enc_ood = None
                self._optim("vae", self.p_vae, 0.0)
            ev_vae = torch.cuda.Event() if par.enabled else None
            if ev_vae is not None:
                ev_vae.record()

        # ---- side branch: the actor forwards + heads, the target cost critics on the N*B rows (beside the VAE phase,
        # where the capped tile loop disturbs the chain least), then the critic phase
        with par.on(0):
            if self._stress_spin and par.enabled and self._stress_at == "head":
                torch.cuda._sleep(self._stress_spin)  # (tests: the side branch arrives late)
            if carried is not None and carried._polyak_pending:
                # (plan.ood_rows in a no-join graph: the cost critics' target update of the previous step -- behind its last
                # reader, the forward on the selected rows at that step's side-branch tail, and behind its main chain's
                # optimizer step, which this branch's fork waited for; in front of this step's readers of the targets: the
                # critic phase on this branch, the cost phase on the main chain through its wait for the action draws)
                m.groups["cost_critic"].polyak_step(m.tau)
                carried._polyak_pending = False
            if carried is not None and carried._actor_pending:
                carried._optim("actor", carried.p_actor, m.tau)
                carried._actor_pending = False
            if carried is not None:
                # (created HERE, behind the main chain's VAE launches: the graph executor keeps the FIRST-created successor
                # of a node on that node's queue -- issued right at the fork, the dual step was the first successor of the
                # previous step's last Adam and the two chains swapped queues at every boundary: 2155 vs 2320 steps/s)
                carried.dual_step()
            if nxt is not None and self.plan.pipe_prologue == "head":
                # (lab: the next step's prologue FIRST on this branch -- covered by ev_critic like "critic", and the N*B-row
                # launch behind it starts ~14 us later against the main chain's VAE launches)
                nxt.prologue(device_noise)
                self._prologue_covered = True
            if self.plan.head_tails:
                # every action draw of the step (cpq.py:141 a_next, :159 a_next2, :164-176 the N OOD draws, :209 the
                # actor-phase sample) by the actor trunks' own forward launch, from its LDS-resident head tiles: four
                # single-purpose launches (30 us on this branch inside the step, profiles/r3_timeline_*.txt) fewer
                hn, ho = self.r_actor_next.forward_with(
                    (self.nobs,), self.r_actor_obs, (self.obs,),
                    tail=G.gauss_tail(ad, m.max_action, eps=nz["eps_next_cc"], a=self.a_next2, eps2=nz["eps_next_c"],
                                      a2=self.a_next),
                    other_tail=G.gauss_tail(ad, m.max_action, eps2=nz["eps_actor"], a2=self.a_pi, tanh2=self.tanh_u,
                                            eps_ood=nz["eps_ood"], n_samples=N, sampled=self.sampled))
                head_next, head_obs = hn[0], ho[0]
                ev_next2 = par.mark(0)
            else:
                hn, ho = self.r_actor_next.forward_with((self.nobs,), self.r_actor_obs, (self.obs,))
                head_next, head_obs = hn[0], ho[0]
                G.gauss_head(head_next, nz["eps_next_cc"], B, ad, m.max_action, a=self.a_next2)
                ev_next2 = par.mark(0)
                G.gauss_head(head_next, nz["eps_next_c"], B, ad, m.max_action, a=self.a_next)
                G.gauss_ood_sample(head_obs, nz["eps_ood"], N, B, ad, self.sampled)
                # the actor-phase sample (cpq.py:209) needs only this forward and its own noise
                G.gauss_head(head_obs, nz["eps_actor"], B, ad, m.max_action, a=self.a_pi, tanh_u=self.tanh_u)
            qc_s = None
            if not self.ood_rows:  # (plan.ood_rows: this forward runs BEHIND the KL quantile, on the rows that pass it)
                self._pr("costold_ood", 0)
                qc_s = self.r_costold_ood.forward(self.obs, self.sampled, map0=L.MAP_MOD, div0=B, share_k16=self.pre_cost)
                self._pr("costold_ood", 1)
            if nxt is not None and self.plan.pipe_prologue == "critic":
                # the next step's prologue HERE: the main chain waits for this branch's critic Adam below (ev_critic), so
                # the next step's first launch needs no edge of its own from this branch
                nxt.prologue(device_noise)
                self._prologue_covered = True
            # critic_loss (cpq.py:137-153)
            self._pr("critic_fwd", 0)
            y_old, q = self.r_old_next.forward_with((self.nobs, self.a_next), self.r_critic, (self.obs, self.act))
            self._pr("critic_fwd", 1)
            if sd is not None:
                self.r_critic.backward_dz(seed=sd["critic"])
            else:
                G.cpq_critic_loss(y_old[:nq], nq, y_old[nq:], nqc, q, nq, self.rew, self.done, B, m.gamma, m.q_thres,
                                  rg, self.dq, st.stat_ptr("loss/critic_loss"))
                self.r_critic.backward_dz()
            if dp is None:
                self._optim("critic", self.p_critic, m.tau)
            else:  # collectives stay on the capture stream (same order on every rank): reduced + stepped over there
                self.p_critic.launch()
                if P.knob("OSRL_DP_SIDE_REDUCE", "1", "DP: the critic group's slab sum on the side branch") == "1":
                    dp.reduce_local(m.groups["critic"])  # the per-rank slab sum is no collective: off the main chain
            ev_critic = par.mark(0)  # also: the side branch's last reader of cost_critic_old is enqueued

        # ---- main: cost_critic_loss (cpq.py:155-201), the part with a gradient: Bellman MSE of the online cost critics
        par.wait(ev_next2)
        qc_old_next, qc = self.r_costold_next.forward_with((self.nobs, self.a_next2), self.r_cost, (self.obs, self.act))
        if sd is not None:
            self.r_cost.backward_dz(seed=sd["cost"])
        else:
            G.cpq_cost_loss(qc_old_next, nqc, qc, nqc, None, self.cost, B, m.gamma, m.qc_thres, m.alpha_lr, rg, 1.0,
                            None, self.dqc, st.stat_ptr("loss/cost_critic_loss"))
            self.r_cost.backward_dz()
        dual_on_side = nxt is not None and PIPE_DUAL_SIDE and par.enabled and not no_join
        ev_cost_stat = None
        if dual_on_side:  # (lab: the Bellman part of the logged cost loss is complete here)
            ev_cost_stat = torch.cuda.Event()
            ev_cost_stat.record()
        fuse_cost = dp is None and self.p_cost.can_fuse_adam() and not self.ood_rows
        if not fuse_cost:
            self.p_cost.launch()
        if self.ood_rows and par.enabled and ev_vae is not None and self.plan.ood_rows_late:
            # (plan.ood_rows_late: without the N*B-row cost-critic launch in its first half the side branch reaches the VAE's
            # Adam + the N*B-row encoder launch ~70 us earlier -- beside THIS dW launch, whose 384 small workgroups then take
            # 70 instead of 25 us, gpurun_out/r6oodrows3; the side branch waits for this point instead of the VAE's dW)
            ev_vae = torch.cuda.Event()
            ev_vae.record()
        par.wait(ev_critic)  # Adam + Polyak of this group rewrites cost_critic_old: after its readers on the side branch
        if fuse_cost:
            self.p_cost.launch_adam(m._lrs["cost_critic"], st.ptr, tau=m.tau)
        elif self.ood_rows:
            # the targets' LAST reader of this step -- the forward on the selected OOD rows -- runs at the end of the side
            # branch: the optimizer step here without its Polyak half, the target update behind that reader (polyak_step)
            m.groups["cost_critic"].adam_step(m._lrs["cost_critic"], st.ptr, tau=m.tau, polyak=False)
        elif dp is None:
            self._update("cost_critic", m.tau)
        else:  # both critic groups' gradients in ONE collective (neither update reads the other's result)
            gc, gcc = m.groups["critic"], m.groups["cost_critic"]
            dp.all_reduce_many_([dp.reduce_local(gc), dp.reduce_local(gcc)])
            gc.adam_step(m._lrs["critic"], st.ptr, tau=m.tau)
            gcc.adam_step(m._lrs["cost_critic"], st.ptr, tau=m.tau)

        # ---- side branch, second half: the OOD statistic with the UPDATED vae
        # (round 4, measured and dropped: letting this launch wait for the cost critics' optimizer step instead -- so that it
        # does not stretch the cost phase it runs beside -- is SLOWER, 2161-2165 vs 2242-2244 steps/s at C2, 2257 vs 2285 at
        # C4: the step is throughput-bound, the N*B-row work has to start as early as its inputs allow)
        with par.on(0):
            if ev_vae is not None:
                par.side[0].wait_event(ev_vae)
            if self._stress_spin and par.enabled and self._stress_at == "second":
                torch.cuda._sleep(self._stress_spin)  # (tests: this branch's second half arrives LATE; results must not move)
            if dp is None and par.enabled and self.plan.vae_adam_side and not self.p_vae.can_fuse_adam():
                self._update("vae", 0.0)  # (engine/plan.py vae_adam_side: off the main chain, in front of its only reader)
                if no_join and nxt is not None and VAE_ADAM_EDGE != "0":
                    # ... of THIS step: the next step's VAE phase, on the main chain, reads it too, and steps that are not
                    # joined need an edge of their own for that
                    self._ev_vae_adam = par.mark(0)
            self._pr("enc_ood", 0)  # bench.py: HIP events around the dominant launch as it runs inside the step
            # (the KL rows of cpq.py:178-182 by the encoder launch itself: OSRL_TAIL_VAE_KL)
            self.r_enc_ood.forward(self.obs, self.sampled, map0=L.MAP_MOD, div0=B, tail=G.vae_kl_tail(Lz, self.kl),
                                   share_k16=self.pre_enc)
            self._pr("enc_ood", 1)
            if no_join and nxt is not None and VAE_WAR_EDGE != "0" and not (self.plan.vae_adam_side and not self.p_vae.can_fuse_adam()):
                self._ev_enc_ood = par.mark(0)  # (the next step's VAE Adam, on the main chain, rewrites what this launch read)
            kl_on_side = dp is not None and par.enabled and DP_SIDE_COLL
            if kl_on_side:
                # the batch-GLOBAL quantile (cpq.py:183 over all world * N * B values): gather, selection and masked mean
                # stay on this branch -- the third collective of the step in issue order on every rank (behind [critic |
                # cost-critic gradients], which the main branch issued above); the main branch never waits for it
                kl_all = dp.all_gather_concat(self.kl)
                dp.quantile_select(kl_all, 0.75, self.quant)
                G.cpq_ood_mean(qc_s, nqc, self.kl, self.quant, N, B, rg, self.ood_mean)
            elif dp is not None:
                ev_kl = par.mark(0)
            elif self.ood_rows:
                # cpq.py:183-184 as a row SET: quantile + ascending list of the rows with KL >= quantile in one
                # single-workgroup launch, the target cost critics on those rows (a quarter of N*B; the grid is sized for
                # all of them, workgroups past the count leave at once), their sum.  The cost critics' target update follows
                # on the MAIN branch behind the join (a main -> side edge this late makes the graph executor serialise the
                # actor phase behind this branch: 500 vs 428 us per step, profiles/r6_ood_rows_timeline_side_polyak.txt).
                if nxt is not None and self.plan.pipe_prologue == "early":
                    nxt.prologue(device_noise)
                G.cpq_ood_select(self.kl, N * B, 0.75, self.quant, self.ood_list, self.ood_count)
                self._pr("costold_ood", 0)
                qc_sel = self.r_costold_ood.forward(self.obs, self.sampled, map0=L.MAP_MOD, div0=B, row_list=self.ood_list,
                                                    n_rows_dev=self.ood_count)
                self._pr("costold_ood", 1)
                G.cpq_ood_sum(qc_sel, nqc, N * B, self.ood_count, 1.0 / (float(N) * float(rg if rg > 0 else B)), self.ood_mean)
            elif N * B <= 32768:  # quantile + masked mean in one single-workgroup launch (keys in registers)
                if nxt is not None and self.plan.pipe_prologue == "early":
                    nxt.prologue(device_noise)
                    self._ev_prologue = par.mark(0)
                G.cpq_ood_stat(qc_s, nqc, self.kl, 0.75, N, B, rg, self.quant, self.ood_mean)
            else:
                G.quantile(self.kl, N * B, 0.75, self.quant)
                G.cpq_ood_mean(qc_s, nqc, self.kl, self.quant, N, B, rg, self.ood_mean)
            if dual_on_side:
                par.side[0].wait_event(ev_cost_stat)
                G.cpq_alpha_step(self.ood_mean, m.qc_thres, m.alpha_lr, 1.0, m.log_alpha, st.stat_ptr("loss/cost_critic_loss"))
            if nxt is not None and self.plan.pipe_prologue == "side":
                nxt.prologue(device_noise)  # (pipelined: the next step's minibatch + noise + tick, off the main chain)

        # ---- main: actor_loss  (cpq.py:203-222): needs the updated critic (side branch: this stream waited for
        # ev_critic above -- a second wait on it is one more graph edge, ~6 us on the chain) and cost critic (here)
        self._pr("actor_phase_fwd", 0)
        y = self.r_pi_q.forward(self.obs, self.a_pi)
        self._pr("actor_phase_fwd", 1)
        if dp is not None and not kl_on_side:
            # (OSRL_DP_SIDE_COLL=0 or no branches: round 4's placement)
            # the batch-GLOBAL quantile (cpq.py:183 over all world * N * B values): the gather is a collective, so it
            # is issued from the capture stream -- here, behind the actor phase's forward launch, which is about when
            # the side branch has its KL rows (profiles/r2_timeline.txt: 397 us vs 405 us); the selection and the
            # masked mean go back to the side branch and run beside the actor phase's backward
            par.wait(ev_kl)
            kl_all = dp.all_gather_concat(self.kl)
            if P.knob("OSRL_DP_SIDE_QUANT", "1", "DP: quantile + masked mean back on the side branch") == "1":
                ev_g = torch.cuda.Event() if par.enabled else None
                if ev_g is not None:
                    ev_g.record()
                with par.on(0):
                    if ev_g is not None:
                        par.side[0].wait_event(ev_g)
                    dp.quantile_select(kl_all, 0.75, self.quant)
                    G.cpq_ood_mean(qc_s, nqc, self.kl, self.quant, N, B, rg, self.ood_mean)
            else:
                dp.quantile_select(kl_all, 0.75, self.quant)
                G.cpq_ood_mean(qc_s, nqc, self.kl, self.quant, N, B, rg, self.ood_mean)
        if sd is not None:
            self.r_pi_q.backward_dz(seed=sd["actor"])
            self.r_actor_obs.backward_dz(seed=sd["head"])
        else:
            G.cpq_actor_loss(y[:nq], nq, y[nq:], nqc, B, m.q_thres, rg, self.dq_pi, st.stat_ptr("loss/actor_loss"))
            self.r_pi_q.backward_dz()
            G.gauss_head_bwd(head_obs, nz["eps_actor"], self.tanh_u, self.r_pi_q.dx, nq, B, ad, m.max_action,
                             self.dhead_actor)
            self.r_actor_obs.backward_dz()
        if dp is None:
            carry = no_join and nxt is not None and (self._ev_prologue is not None or self._prologue_covered)
            if self._ev_vae_adam is not None and (VAE_ADAM_EDGE == "actor" or not carry):
                par.wait(self._ev_vae_adam)
                self._ev_vae_adam = None
            if self._ev_enc_ood is not None and (VAE_WAR_EDGE == "actor" or not carry):
                par.wait(self._ev_enc_ood)
                self._ev_enc_ood = None
            if carry and PIPE_ACTOR_SIDE:
                self._actor_pending = True  # (the next step issues it on its side branch, behind its fork)
            else:
                self._optim("actor", self.p_actor, m.tau)
            if carry:
                self._dual_pending = True  # (the next step of this graph runs it: dual_step())
                self._polyak_pending = bool(self.ood_rows)
                return
            par.join(0)
        else:  # actor gradient, the per-rank partial statistics and the partial qc_ood mean in one collective
            self.p_actor.launch()
            par.join(0)
            ga = m.groups["actor"]
            dp.all_reduce_many_([dp.reduce_local(ga), st.stats, self.ood_mean])
            ga.adam_step(m._lrs["actor"], st.ptr, tau=m.tau)
        # dual step + the OOD term of the logged loss (cpq.py:186-195): after the join, so that the side branch has no
        # incoming edge from the main branch after the VAE's Adam (the graph executor keeps two linear chains); under
        # data parallelism the statistics are already the global ones here, so the global term is added once
        if self.ood_rows:  # the cost critics' target update: behind its last reader (side branch, joined above)
            m.groups["cost_critic"].polyak_step(m.tau)
        if not dual_on_side:
            self.dual_step()
        if nxt is not None and self.plan.pipe_prologue == "main":
            nxt.prologue(device_noise)  # (lab: the pipelined graph with the next prologue on the main chain, as a control)

    def dual_step(self) -> None:
        """log_alpha's ascent + the OOD term of this step's logged cost loss (cpq.py:186-195), on the current stream."""
        m = self.model
        G.cpq_alpha_step(self.ood_mean, m.qc_thres, m.alpha_lr, 1.0, m.log_alpha, self.st.stat_ptr("loss/cost_critic_loss"))
        self._dual_pending = False

    # ------------------------------------------------------------------ #
    def load_batch(self, observations, next_observations, actions, rewards, costs, done) -> None:
        load_into(((self.obs, observations), (self.nobs, next_observations), (self.act, actions),
                   (self.rew, rewards), (self.cost, costs), (self.done, done)))

    def load_noise(self, noise: Dict[str, torch.Tensor]) -> None:
        for k in NOISE_KEYS:
            self.noise[k].copy_(torch.as_tensor(noise[k]).reshape(self.noise[k].shape), non_blocking=True)

    def capture(self) -> None:
        """Capture one step into a hipGraph -- on one GPU a few times over, keeping the graph whose replays are fastest
        (core.pick_fastest: the branch -> hardware-queue mapping of a capture depends on the streams created before it)."""
        from .core import CAPTURE_TRIES, pick_fastest

        def replay(c):
            c[0].replay()

        tries = CAPTURE_TRIES if (self.dist is None and self.parallel_branches) else 1
        (self.graph, self._par, self._arena), self.capture_ms = pick_fastest(self._capture_once, replay, self._snapshot,
                                                                              self._restore, tries)

    def _capture_once(self):
        """Capture one step (device-drawn noise) into a hipGraph.  Warm-up launches run first on a
        side stream as torch requires; the model state they advance is restored afterwards."""
        snap = self._snapshot()
        # data parallel: the side branch holds no collective (the critic group's all-reduce + Adam run on the
        # capture stream after the join), so every rank issues its RCCL calls in the same order on one stream
        par = Branches(self.parallel_branches, 2 if self.dist is not None else 1)
        try:  # the warm-up pass and the capture pass both advance the model: ALWAYS put the snapshot back, also
            # when the capture is refused and the caller falls back to eager launches
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            arena = ArgArena(self.dev)  # the fused-MLP launches' descriptors live in HBM (core.ArgArena)
            with torch.cuda.stream(s), arena.record():
                self.body(True, par)
            torch.cuda.current_stream().wait_stream(s)
            arena.upload()
            g = torch.cuda.CUDAGraph()
            # (capturing on a high-priority stream to favour the critical chain halves the throughput: measured 980
            # vs 1755 steps/s -- every kernel of the step ran ~2x slower)
            with graph_capture(g), arena.replay():
                self.body(True, par)
        finally:
            torch.cuda.synchronize()
            self._restore(snap)
        return g, par, arena  # (the side streams and the argument blocks its kernels read stay alive with the graph)

    def _snapshot(self):
        m = self.model
        snap = {"log_alpha": m.log_alpha.clone(), "state": self.st.state.clone(), "host": self.st.host_step,
                "stats": self.st.stats.clone(), "ring": self.st.ring.clone()}
        for n, g in m.groups.items():
            snap[n] = (g.p.clone(), g.m.clone(), g.v.clone(), None if g.tgt is None else g.tgt.clone())
        return snap

    def _restore(self, snap) -> None:
        m = self.model
        m.log_alpha.copy_(snap["log_alpha"])
        self.st.state.copy_(snap["state"])
        self.st.stats.copy_(snap["stats"])
        self.st.ring.copy_(snap["ring"])
        self.st.host_step = snap["host"]
        for n, g in m.groups.items():
            p, mm, v, t = snap[n]
            g.p.copy_(p)
            g.m.copy_(mm)
            g.v.copy_(v)
            if t is not None:
                g.tgt.copy_(t)
        m.repack()

    def attach_replay(self, store) -> None:
        """Sample minibatches on device from ``store`` (common/replay.py) inside the step itself."""
        self.replay = store
        self.graph = None
        self._pipe = None

    def _run(self, use_graph: bool) -> None:
        """Replay the captured step (capturing it first).  With a DataParallel hook the RCCL collectives
        are captured into the same hipGraph; if the runtime refuses (older RCCL), fall back to eager."""
        if self.dist is not None and os.environ.get("OSRL_DP_EAGER") == "1":
            use_graph = False  # operator override: run the data-parallel step without capturing its collectives
        if use_graph and not self._graph_failed:
            if self.graph is None:
                ok = True
                try:
                    self.capture()
                except Exception as e:  # pragma: no cover - depends on the RCCL build
                    if self.dist is None:
                        raise
                    import warnings
                    warnings.warn(f"hipGraph capture of the data-parallel step failed ({e!r}); running eagerly")
                    ok = False
                if self.dist is not None and not self.dist.all_agree(ok, self.dev):
                    self._graph_failed, self.graph = True, None  # every rank runs eagerly, or none does
            if self.graph is not None:
                self.graph.replay()
                self.st.host_step += 1
                return
        self.body(True)

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
        self._run(use_graph)

    def step(self, observations, next_observations, actions, rewards, costs, done, noise=None,
             use_graph: bool = True) -> None:
        check_plans_current(self)
        if self.replay is not None:
            raise RuntimeError("a replay store is attached: call step_replay() (or attach_replay(None))")
        self.load_batch(observations, next_observations, actions, rewards, costs, done)
        if noise is not None:
            self.load_noise(noise)
            self.body(False)
            return
        self._run(use_graph)
