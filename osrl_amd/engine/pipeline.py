"""Several train steps per hipGraph, software-pipelined ACROSS steps.

Inside one step every rearrangement has been measured out (DESIGN_LOG rounds 3-5: third branches serialise, the graph
executor runs two linear chains and nothing more).  What one step cannot use are the stretches where a queue idles
because of the STEP BOUNDARY: BCQ-Lag's side queue during the actor phase (C3: 1318-1540 us) while the next step's VAE
phase then runs alone on the main queue (0-205 us); CPQ's ~20 us of prologue / join / dual step at head and tail.  Step
k+1's prologue and VAE phase depend only on (a) minibatch and noise k+1 and (b) the VAE after step k's VAE optimizer step,
and nothing reads the VAE after step k's last N*B-row decoder / encoder launch -- so with

  * a TWIN engine (same model, same flat groups, same plans; its own batch / noise / activation buffers) for the odd steps,
  * two device step states that take turns (``StepState.link``: bias corrections and Philox offsets explicit per state;
    osrl_step_begin_peer gives each tick max(own, peer) + 1),

the graph of n steps issues step k+1's head on the side queue under step k's tail.  Nothing changes numerically: the
same kernels run on the same inputs in the same order per step -- parameters after n pipelined steps are BIT-EQUAL to n
replays of the one-step graph (tests/test_gpu_pipeline.py).  The overlap exists only inside a graph (n - 1 of n
boundaries); a replay boundary is a full join, so an engine's state between replays is that of exactly n more steps.

Reference: one step = CPQTrainer.train_one_step (osrl/algorithms/cpq.py:294-313) / BCQLTrainer.train_one_step
(bcql.py:283-306) on a minibatch of TransitionDataset (dataset.py:832-847); the reference runs them strictly one after the
other from a Python loop (examples/train/train_cpq.py:138-144).
"""
from __future__ import annotations

from typing import Optional

import torch

from .core import ArgArena, Branches, graph_capture, slab_epochs, check_plans_current

STEPS_PER_GRAPH = 4  # (default of PipelinedSteps(engine); engines take their plan's steps_per_graph, engine/plan.py)

class PipelinedSteps:
    """``run(n)``: n train steps of ``engine`` (CPQ or BCQ-Lag, single GPU, replay store attached) on minibatches drawn
    on device -- whole multiples of ``steps_per_graph`` through the pipelined graph, the remainder through the engine's
    one-step graph."""

    def __init__(self, engine, steps_per_graph: Optional[int] = None):
        if getattr(engine, "dist", None) is not None:
            raise RuntimeError("pipelined steps are a single-GPU plan (a data-parallel step keeps its collectives on "
                               "the capture stream in one fixed order)")
        if engine.replay is None:
            raise RuntimeError("pipelined steps draw their minibatches on device: attach a replay store first")
        self.n = int(steps_per_graph or STEPS_PER_GRAPH)
        if self.n < 2:
            raise ValueError("steps_per_graph >= 2")
        m = engine.model
        e0 = engine
        # the twin: same model (parameters, moments, targets, gradient slabs are the MODEL's flat groups), same plans,
        # same seed (the Philox keys of a step depend on the step count, not on which engine runs it)
        e1 = type(engine)(m, engine.B, rows_global=engine.rows_global, seed=engine.seed)
        e1.attach_replay(engine.replay)
        # building the twin re-attached (identical) plans to the groups' slabs: both engines are current
        e0._slab_epochs = e1._slab_epochs = slab_epochs(m)
        e0.st.link(e1.st)
        e0.graph = None  # (its one-step graph was captured with an unlinked state: the tick must look at the peer now)
        self.e = (e0, e1)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._par = self._arena = None

    # ------------------------------------------------------------------ #
    def _issue(self, par: Branches) -> None:
        """The n steps as they are captured: step k on engine k % 2, with step k+1's head issued from inside step k."""
        E, n = self.e, self.n
        for k in range(n):
            e, x = E[k % 2], (E[(k + 1) % 2] if k + 1 < n else None)
            if hasattr(e, "head"):
                e.body(True, par, nxt=x, head_done=k > 0)
            else:
                e.body(True, par, nxt=x, prologue_done=k > 0, prev=E[(k - 1) % 2] if k > 0 else None)

    def _snapshot(self):
        e0, e1 = self.e
        return e0._snapshot(), (e1.st.state.clone(), e1.st.stats.clone())

    def _restore(self, snap) -> None:
        e0, e1 = self.e
        e0._restore(snap[0])
        e1.st.state.copy_(snap[1][0])
        e1.st.stats.copy_(snap[1][1])

    def capture(self) -> None:
        """(A few captures, the fastest graph kept: core.pick_fastest.)"""
        from .core import CAPTURE_TRIES, pick_fastest
        (self.graph, self._par, self._arena), self.capture_ms = pick_fastest(self._capture_once, lambda c: c[0].replay(),
                                                                              self._snapshot, self._restore, CAPTURE_TRIES)

    def _capture_once(self):
        e0, e1 = self.e
        dev = e0.st.state.device
        snap = e0._snapshot()
        snap1 = (e1.st.state.clone(), e1.st.stats.clone())
        par = Branches(True, 1)
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            arena = ArgArena(dev, capacity=1 << 20)
            with torch.cuda.stream(s), arena.record():
                self._issue(par)
            torch.cuda.current_stream().wait_stream(s)
            arena.upload()
            g = torch.cuda.CUDAGraph()
            with graph_capture(g), arena.replay():
                self._issue(par)
        finally:
            torch.cuda.synchronize()
            e0._restore(snap)
            e1.st.state.copy_(snap1[0])
            e1.st.stats.copy_(snap1[1])
        return g, par, arena

    def run(self, n_steps: int) -> None:
        e0 = self.e[0]
        check_plans_current(e0)
        check_plans_current(self.e[1])
        q, r = divmod(int(n_steps), self.n)
        if q:
            if self.graph is None:
                self.capture()
            for _ in range(q):
                self.graph.replay()
                e0.st.host_step += self.n
        for _ in range(r):
            e0.step_replay(True)
