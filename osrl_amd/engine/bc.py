"""The BC train step (osrl/algorithms/bc.py:45-52,103-109): ONE launch (csrc/mlp.hip ``mlp_step_kernel``) where the
shape allows it, otherwise the static plan of six launches; either way replayed as a hipGraph."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from .. import _lib as L
from . import plan as _plan
from ..common.net import net_desc_seq
from . import glue as G
from .core import ArgArena, DwPlan, MlpRun, StepState, capture_step, cur_stream, load_into, check_plans_current

STAT_KEYS = ["loss/actor_loss"]


class BCEngine:
    def __init__(self, model, batch_size: int, rows_global: int = 0, dist=None):
        m = self.model = model
        B = self.B = int(batch_size)
        self.rows_global, self.dist = int(rows_global), dist
        dev = torch.device(m.device)
        f = dict(dtype=torch.float32, device=dev)
        self.st = StepState(dev, STAT_KEYS)
        self.obs = torch.zeros(B, m.actor.pi[0].in_features, **f)
        self.act = torch.zeros(B, m.action_dim, **f)
        self.d_pi = net_desc_seq([m.actor.pi], float(m.max_action))
        m.repack()
        self.r_pi = MlpRun(self.d_pi, B, True, dev)
        self.du = torch.zeros(1, B, m.action_dim, **f)
        self.r_pi.setup_backward(self.du)
        self.plan = DwPlan(m.groups["actor"], self.r_pi.dw_entries(), B, dev)
        # every dW plan of this engine is built: the slab epochs they were built against are recorded NOW (not at the
        # first step), so an engine that is constructed directly, never stepped and then superseded is flagged stale
        from .core import slab_epochs
        self._slab_epochs = slab_epochs(self.model)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.replay = None
        # gather -> forward -> MSE -> backward -> dW -> Adam -> tick as one launch (include/osrl_amd.h
        # osrl_mlp_regress_step): single device, one row split per dW tile (no gradient slabs to sum), a work list and
        # a row-tile count that fit the resident grid; the library has the last word (OSRL_E_UNSUPPORTED -> the plan)
        self.one_launch = (_plan.knob("OSRL_BC_ONE_LAUNCH", "1", "the BC step as one launch", operator=True) == "1" and dist is None and self.plan.tile_blocks == 4
                           and self.plan.n_splits == 1 and 0 < self.plan.n_work <= L.STEP_MAX_WG
                           and B <= 16 * L.STEP_MAX_WG)
        self.step_ws = torch.zeros(L.STEP_WS, **f) if self.one_launch else None
        if self.one_launch:
            self.st.health_checks.append(self.check_health)
        self._step_c = None
        # A one-kernel step is launched directly: replaying a one-node hipGraph costs 42.4 us per step where the
        # launch itself costs 39.2 (tools/bc_eager_vs_graph.py).  Its descriptor still lives in HBM (ArgArena:
        # recorded by the first step, looked up by the later ones), so a box that keeps kernel arguments in host
        # memory does not fetch 3 KB per wave over PCIe.  OSRL_BC_DIRECT=0: capture it like every other step.
        self.direct = _plan.knob("OSRL_BC_DIRECT", "1", "the one-launch BC step launched directly (no graph)") == "1"
        self._arena_direct: Optional[ArgArena] = None
        # the one-launch step's own dW work list: 32 x 32 tiles when they fit the resident grid (a 256-row batch gives a
        # 64 x 64 tile 3.4 us of MFMA work on ONE CU; a quarter of that on four CUs), else the plan's 64 x 64 list
        self._step_work, self._step_T = self.plan.d_work, 4
        if self.one_launch:
            work = []
            for i, (_dz, _a, wk, _bk) in enumerate(self.r_pi.dw_entries()):
                out_f, in_f = m.groups["actor"].layout[wk][1]
                work += [v for ot in range((out_f + 31) // 32) for it in range((in_f + 31) // 32)
                         for v in (i, ot, it, 0 | (1 << 16))]
            if len(work) // 4 <= L.STEP_MAX_WG and _plan.knob("OSRL_BC_STEP_T", "2", "dW tile of the one-launch BC step (16-blocks)") == "2":
                self._step_work, self._step_T = torch.tensor(work, dtype=torch.int32, device=dev), 2

    def _one_launch_args(self) -> "L.MlpStepT":
        """The descriptor of osrl_mlp_regress_step for the current replay store (static: built once per attachment)."""
        if self._step_c is not None:
            return self._step_c
        m, st, grp, r = self.model, self.st, self.model.groups["actor"], self.r_pi
        k = L.MlpStepT()
        k.st, k.beta1, k.beta2, k.warmup = st.ptr, st.betas[0], st.betas[1], st.warmup
        k.n_stats, k.ring_len = st.n_stats, st.ring_len
        k.stats_cur, k.ring = st.stats.data_ptr(), st.ring.data_ptr()
        if self.replay is not None:
            n_f, src, dst, w, sc, n_rows, _B, g_seed, g_stream, keep = self.replay.gather_args((self.obs, self.act), (0, 2))
            k.n_fields, k.n_rows, k.gather_seed, k.gather_stream = n_f, n_rows, g_seed, g_stream
            for i in range(n_f):
                k.src[i], k.dst[i], k.width[i], k.scale[i] = src[i], dst[i], w[i], sc[i]
            self._step_keep = keep
        k.net = r.fwd_c
        k.in_ = r._rows(self.obs)
        k.acts = r.acts_c
        k.grads = r.grads_c
        k.target = self.act.data_ptr()
        k.n_global = (self.rows_global or self.B) * m.action_dim
        k.stat = st.stat_ptr("loss/actor_loss")
        k.entries, k.work = self.plan.d_entries.data_ptr(), self._step_work.data_ptr()
        k.n_work, k.tile_blocks = self._step_work.numel() // 4, self._step_T
        k.p, k.m, k.v = grp.p.data_ptr(), grp.m.data_ptr(), grp.v.data_ptr()
        k.map_f, k.map_b, k.pf, k.pb = grp._map_f.data_ptr(), grp._map_b.data_ptr(), grp.pf.data_ptr(), grp.pb.data_ptr()
        k.lr, k.eps = m._lrs["actor"], 1e-8
        k.ws = self.step_ws.data_ptr()
        self._step_c = k
        return k

    def one_launch_failed(self) -> bool:
        """True when a workgroup of the one-launch step gave up waiting for the row tiles (synchronises; never seen:
        all its workgroups are resident at once).  The parameters are invalid after that."""
        return bool(self.one_launch and self.step_ws[L.STEP_MAX_WG + 2].item() != 0)

    def check_health(self) -> None:
        """Runs wherever the host synchronises with the step anyway (StepState.health_checks: statistics flush,
        device_step; BCTrainer.evaluate; checkpoint save).  The one-launch step's dW workgroups WAIT for the row tiles
        with a bounded poll; on a device where they are not all resident at once (a partitioned or CU-masked GPU) the
        poll can expire -- the kernel then flags it and goes on with an incomplete gradient.  That must not pass
        silently: the flag is cleared, the engine falls back to the six-launch plan for good, and the caller is told
        that the parameters since the last good checkpoint are invalid."""
        if self.one_launch and self.step_ws[L.STEP_MAX_WG + 2].item() != 0:
            self.one_launch = False
            self._arena_direct = None
            self.step_ws.zero_()
            raise RuntimeError(
                "osrl_amd: a workgroup of the one-launch BC step gave up waiting for the row tiles (not all of the "
                "launch's workgroups were resident at once on this device); parameters updated since the last "
                "statistics read are INVALID -- restore a checkpoint.  The engine now runs the six-launch plan "
                "(OSRL_BC_ONE_LAUNCH=0 selects it from the start).")

    def attach_replay(self, store) -> None:
        """Sample (observations, actions) minibatches on device from ``store`` (common/replay.py) inside the step:
        TransitionDataset + DataLoader + H2D of train_bc.py:105-121 folded into the captured graph."""
        if store is not None and store.widths[0] != self.obs.shape[1]:
            raise ValueError(f"the store's observations have {store.widths[0]} columns, the policy reads "
                             f"{self.obs.shape[1]} (bc_mode='multi-task' appends the cost return: process_bc_dataset)")
        self.replay = store
        self.graph = None
        self._step_c = None
        self._arena_direct = None

    def _run(self, use_graph: bool) -> None:
        check_plans_current(self)  # (also before the replay of an already captured graph)
        if use_graph and self.dist is None and self.one_launch and self.direct:
            if self._arena_direct is None:  # this step records the launch's descriptor; the later ones read it from HBM
                arena = ArgArena(self.st.state.device, capacity=1 << 14)
                with arena.record():
                    self.body()
                arena.upload()
                # (if the library refused the shape, body() ran the six launches and cleared one_launch: this step is
                # done either way, and the next one takes the captured plan below)
                self._arena_direct = arena if self.one_launch else None
            else:
                with self._arena_direct.replay():
                    self.body()
            return
        if use_graph and self.dist is None:
            if self.graph is None:
                self._capture()
            self.graph.replay()
            self.st.host_step += 1
        else:
            self.body()

    def step_replay(self, use_graph: bool = True) -> None:
        assert self.replay is not None
        self._run(use_graph)

    def body(self) -> None:
        check_plans_current(self)
        m, B, ad = self.model, self.B, self.model.action_dim
        if self.one_launch:
            rc = L.load().osrl_mlp_regress_step(C.byref(self._one_launch_args()), cur_stream())
            if rc == L.E_UNSUPPORTED:
                self.one_launch = False  # (a shape outside the fused launch's: the plan below, from now on)
            else:
                L.check(rc, "osrl_mlp_regress_step")
                self.st.host_step += 1
                return
        self.st.prologue(self.replay, (self.obs, self.act), None, 0, False, fields=(0, 2))
        pred = self.r_pi.forward(self.obs)[0]
        ng = (self.rows_global or B) * ad
        G.mse_loss(pred, self.act, B * ad, ng, self.du, self.st.stat_ptr("loss/actor_loss"))
        self.r_pi.backward_dz()
        self.plan.launch()
        grp = m.groups["actor"]
        if self.dist is not None:
            self.dist.allreduce_group(grp)
        grp.adam_step(m._lrs["actor"], self.st.ptr)
        if self.dist is not None:  # per-rank partial of the globally normalised loss -> the global value
            self.dist.all_reduce_(self.st.stats)

    def step(self, observations, actions, use_graph: bool = True) -> None:
        if self.replay is not None:
            raise RuntimeError("a replay store is attached: call step_replay() (or attach_replay(None))")
        load_into(((self.obs, observations), (self.act, actions)))
        self._run(use_graph)

    def _capture(self) -> None:
        check_plans_current(self)
        g = self.model.groups["actor"]
        snap = (g.p.clone(), g.m.clone(), g.v.clone(), self.st.state.clone(), self.st.stats.clone(),
                self.st.ring.clone(), self.st.host_step)
        gr, self._arena = capture_step(self.st.state.device, self.body, self.body)
        torch.cuda.synchronize()
        g.p.copy_(snap[0]); g.m.copy_(snap[1]); g.v.copy_(snap[2])
        self.st.state.copy_(snap[3]); self.st.stats.copy_(snap[4]); self.st.ring.copy_(snap[5])
        self.st.host_step = snap[6]
        self.model.repack()
        self.graph = gr
