"""The BC train step as a static launch plan (osrl/algorithms/bc.py:45-52,103-109)."""
from __future__ import annotations

from typing import Optional

import torch

from ..common.net import net_desc_seq
from . import glue as G
from .core import DwPlan, MlpRun, StepState, capture_step, load_into

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
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.replay = None

    def attach_replay(self, store) -> None:
        """Sample (observations, actions) minibatches on device from ``store`` (common/replay.py) inside the step:
        TransitionDataset + DataLoader + H2D of train_bc.py:105-121 folded into the captured graph."""
        if store is not None and store.widths[0] != self.obs.shape[1]:
            raise ValueError(f"the store's observations have {store.widths[0]} columns, the policy reads "
                             f"{self.obs.shape[1]} (bc_mode='multi-task' appends the cost return: process_bc_dataset)")
        self.replay = store
        self.graph = None

    def step_replay(self, use_graph: bool = True) -> None:
        assert self.replay is not None
        if use_graph and self.dist is None:
            if self.graph is None:
                self._capture()
            self.graph.replay()
            self.st.host_step += 1
        else:
            self.body()

    def body(self) -> None:
        m, B, ad = self.model, self.B, self.model.action_dim
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
        if use_graph and self.dist is None:
            if self.graph is None:
                self._capture()
            self.graph.replay()
            self.st.host_step += 1
        else:
            self.body()

    def _capture(self) -> None:
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
