"""Several train steps per hipGraph, software-pipelined across steps (osrl_amd/engine/pipeline.py).

The contract: n pipelined steps == n replays of the one-step graph, BIT for bit -- parameters, Polyak targets, Adam
moments, ``log_alpha`` / PID state, every logged statistic of every step, the device step count.  (Reference: the steps
are CPQTrainer.train_one_step cpq.py:294-313 / BCQLTrainer.train_one_step bcql.py:283-306 on TransitionDataset
minibatches, dataset.py:832-847; the reference runs them one after the other from a Python loop.)  Oracle parity of the
pipelined path itself: the replayed 2-step graph's batches and noise are read back and handed to the pinned oracle.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cases import CASES  # noqa: E402
from gpu_util import build_gpu  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _small(name, n_store=4096):
    """A small golden case with a device-resident replay store attached to its engine."""
    from osrl_amd.common.replay import ReplayStore, synthetic_transitions
    c = CASES[name]
    m, tr, lg = build_gpu(c, stats_mode="none", use_graph=True)
    eng = m.engine(c.B)
    store = ReplayStore(synthetic_transitions(n_store, c.od, c.ad, seed=7, max_action=c.max_action), torch.device(DEV),
                        reward_scale=0.1, cost_scale=1.0, seed=3)
    eng.attach_replay(store)
    return m, eng


def _bench(name):
    import bench
    wl = bench.Workload(name, torch.device(DEV), 0, 1, None, n_store=1 << 16, use_graph=True)
    return wl.model, wl.eng


def _state(m, eng):
    out = {}
    for n, g in m.groups.items():
        out[n + ".p"], out[n + ".m"], out[n + ".v"] = g.p.clone(), g.m.clone(), g.v.clone()
        if g.tgt is not None:
            out[n + ".tgt"] = g.tgt.clone()
        if g.pf is not None:
            out[n + ".pf"], out[n + ".pb"] = g.pf.clone(), g.pb.clone()
    for k in ("log_alpha", "pid_state"):
        if isinstance(getattr(m, k, None), torch.Tensor):
            out[k] = getattr(m, k).clone()
    return out


@pytest.mark.parametrize("name,spg,total", [
    ("cpq_small", 2, 6), ("cpq_small", 4, 9), ("cpq_odd", 3, 7), ("cpq_wide", 2, 4),
    ("bcql_small", 2, 6), ("bcql_small", 4, 9), ("bcql_pid", 3, 7), ("bcql_wide", 2, 4),
    ("c2", 2, 4), ("c2", 4, 8), ("c4", 4, 4), ("c3", 2, 4)])
def test_pipelined_steps_equal_one_step_graph_replays(name, spg, total):
    """``PipelinedSteps.run(total)`` (whole graphs of ``spg`` steps + single-step remainder) against ``total`` replays of
    the one-step graph from the same initial state: every flat group's parameters / moments / targets / packed copies,
    the dual variable / PID integrators, the statistics of EVERY step (two of them still uncommitted in the two step
    states' own buffers, the rest in the shared ring) and the step count -- bit-equal."""
    from osrl_amd.engine.pipeline import PipelinedSteps
    build = _bench if name in ("c2", "c3", "c4") else _small
    m_a, e_a = build(name)
    for _ in range(total):
        e_a.step_replay(True)
    torch.cuda.synchronize()
    assert e_a.graph is not None
    ref = _state(m_a, e_a)
    ref_stats = [e_a.st.read_stats(s) for s in range(1, total + 1)]
    assert e_a.st.device_step() == total
    del m_a, e_a
    torch.cuda.empty_cache()

    m_b, e_b = build(name)
    e_b.steps_replay(total, steps_per_graph=spg)  # (the engine-level entry point; builds its PipelinedSteps on first use)
    torch.cuda.synchronize()
    pipe = e_b._pipe
    assert isinstance(pipe, PipelinedSteps) and pipe.n == spg
    assert pipe.graph is not None and e_b.st.device_step() == total and e_b.st.host_step == total
    got = _state(m_b, e_b)
    assert set(got) == set(ref)
    for k in ref:
        assert torch.equal(ref[k], got[k]), f"{name} spg={spg}: {k} differs after {total} steps " \
                                             f"(max |d| = {(ref[k] - got[k]).abs().max().item():.3e})"
    for s in range(1, total + 1):
        st = e_b.st.read_stats(s)
        for k, v in ref_stats[s - 1].items():
            assert st[k] == v or (np.isnan(st[k]) and np.isnan(v)), f"{name} spg={spg}: statistic {k} of step {s}: {st[k]} vs {v}"
    many = e_b.st.read_stats_many(range(1, total + 1))
    for s in range(1, total + 1):
        assert many[s] == [ref_stats[s - 1][k] for k in e_b.st.keys], f"read_stats_many, step {s}"
    # ... and the engine keeps counting correctly when single steps follow the pipelined ones
    e_b.step_replay(True)
    torch.cuda.synchronize()
    assert e_b.st.device_step() == total + 1
    assert all(np.isfinite(v) for v in e_b.st.read_stats().values())


@pytest.mark.parametrize("prologue", ["early", "critic", "head"])
@pytest.mark.parametrize("name,spg,total", [("cpq_small", 2, 6), ("cpq_small", 4, 9), ("cpq_odd", 3, 7), ("cpq_wide", 2, 4),
                                             ("c2", 4, 8), ("c4", 4, 8)])
def test_unjoined_pipelined_steps_equal_one_step_graph_replays(name, spg, total, prologue, monkeypatch):
    """The no-join form of a pipelined CPQ graph (plan.pipe_no_join; C4's pinned plan; forced here on every case through the
    lab switches): the steps of a graph are not joined, step k's dual step (cpq.py:186-195) is issued at the head of step
    k+1's side branch, and the next prologue sits either in front of the OOD statistic (an event of its own for the main
    chain), in front of the critic phase or first on the side branch (both covered by the wait for the critic's Adam).  Same contract as above: parameters,
    moments, targets, ``log_alpha``, EVERY step's statistics (the cost loss's OOD term is added one step later on another
    queue) and the step count are bit-equal to replays of the one-step graph."""
    from osrl_amd.engine.pipeline import PipelinedSteps
    build = _bench if name in ("c2", "c4") else _small
    m_a, e_a = build(name)
    for _ in range(total):
        e_a.step_replay(True)
    torch.cuda.synchronize()
    ref = _state(m_a, e_a)
    ref_stats = [e_a.st.read_stats(s) for s in range(1, total + 1)]
    del m_a, e_a
    torch.cuda.empty_cache()

    monkeypatch.setenv("OSRL_LAB", "1")
    monkeypatch.setenv("OSRL_PIPE_DUAL", "next")
    monkeypatch.setenv("OSRL_PIPE_PROLOGUE", prologue)
    m_b, e_b = build(name)
    assert e_b.plan.pipe_no_join and e_b.plan.pipe_prologue == prologue
    e_b.steps_replay(total, steps_per_graph=spg)
    torch.cuda.synchronize()
    pipe = e_b._pipe
    assert isinstance(pipe, PipelinedSteps) and pipe.n == spg and pipe.e[1].plan == e_b.plan
    assert not pipe.e[0]._dual_pending and not pipe.e[1]._dual_pending, "a graph's last step runs its own dual step"
    assert e_b.st.device_step() == total
    got = _state(m_b, e_b)
    for k in ref:
        assert torch.equal(ref[k], got[k]), f"{name} spg={spg} no-join/{prologue}: {k} differs after {total} steps " \
                                             f"(max |d| = {(ref[k] - got[k]).abs().max().item():.3e})"
    for s in range(1, total + 1):
        st = e_b.st.read_stats(s)
        for k, v in ref_stats[s - 1].items():
            assert st[k] == v or (np.isnan(st[k]) and np.isnan(v)), f"{name} no-join/{prologue}: statistic {k} of step {s}: {st[k]} vs {v}"
    many = e_b.st.read_stats_many(range(1, total + 1))
    for s in range(1, total + 1):
        assert many[s] == [ref_stats[s - 1][k] for k in e_b.st.keys], f"read_stats_many, step {s}"


def _spin_cycles(us):
    """Cycles of torch.cuda._sleep's spin kernel for ~``us`` microseconds on this device (calibrated here)."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    a.record()
    torch.cuda._sleep(2_000_000)
    b.record()
    torch.cuda.synchronize()
    return max(1000, int(2_000_000 * us / (a.elapsed_time(b) * 1e3)))


@pytest.mark.parametrize("name,edge,at", [("c2", "OSRL_VAE_ADAM_EDGE", "second"), ("c4", "OSRL_VAE_WAR_EDGE", "second"),
                                          ("c2", None, "head"), ("c4", None, "head"), ("c2", None, "main"), ("c4", None, "main")])
def test_unjoined_graphs_are_ordered_by_edges_not_by_timing(name, edge, at, monkeypatch):
    """The steps of a no-join graph overlap across their boundary, so every cross-queue dependency between step k's side
    branch and step k+1's main chain needs a graph EDGE -- two of them had none until round 6's third session and were
    ordered by ~100-280 us of timing slack: C2's VAE Adam on the side branch -> the next step's VAE phase (RAW on the VAE's
    weights, WAR on its gradient slabs), and C4's N*B-row encoder launch at the tail of the side branch -> the next step's
    VAE Adam on the main chain (WAR on the weights).  Here the side branch's second half of EVERY step is held back by a
    ~600 us spin kernel (longer than a step): with the edges the pipelined graph still gives the one-step graph's bits --
    parameters, moments, targets, ``log_alpha``, every step's statistics -- and, as the control that the delay reaches the
    hazard, the same graph built WITHOUT the edge (lab switch) does not.  The same delay at the head of the side branch and
    at the head of the main chain (no hazard known there, no control): same bits."""
    m_a, e_a = _bench(name)
    total, spg = 6, 3
    for _ in range(total):
        e_a.step_replay(True)
    torch.cuda.synchronize()
    ref = _state(m_a, e_a)
    ref_stats = [e_a.st.read_stats(s) for s in range(1, total + 1)]
    del m_a, e_a
    torch.cuda.empty_cache()
    monkeypatch.setenv("OSRL_LAB", "1")
    monkeypatch.setenv("OSRL_STRESS_SPIN_CYCLES", str(_spin_cycles(600.0)))
    monkeypatch.setenv("OSRL_STRESS_SPIN_AT", at)

    def run():
        m_b, e_b = _bench(name)
        assert e_b.plan.pipe_no_join and e_b._stress_spin > 0
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        e_b.steps_replay(total, steps_per_graph=spg)  # (capture + first replays)
        torch.cuda.synchronize()
        got, stats = _state(m_b, e_b), [e_b.st.read_stats(s) for s in range(1, total + 1)]
        t0.record()
        e_b.steps_replay(spg, steps_per_graph=spg)
        t1.record()
        torch.cuda.synchronize()
        return got, stats, t0.elapsed_time(t1) * 1e3 / spg

    got, stats, us = run()
    assert us > 500.0, f"the spin kernel is not in the graph ({us:.0f} us per step)"
    for k in ref:
        assert torch.equal(ref[k], got[k]), f"{name}: {k} moves when the side branch is late (max |d| = {(ref[k] - got[k]).abs().max().item():.3e})"
    for s in range(total):
        for k, v in ref_stats[s].items():
            assert stats[s][k] == v, f"{name}: statistic {k} of step {s + 1} moves when the side branch is late: {stats[s][k]} vs {v}"
    if edge is None or os.environ.get("OSRL_FUSE_DW_ADAM", "auto") != "auto":
        return  # (with the fused dW + Adam launches forced, the VAE's Adam sits on the main chain: no hazard to reach at C2)
    # control: without the edge the same delay changes the result
    monkeypatch.setenv(edge, "0")
    got0, stats0, _ = run()
    same = all(torch.equal(ref[k], got0[k]) for k in ref) and all(stats0[s][k] == v for s in range(total) for k, v in ref_stats[s].items())
    assert not same, f"{name}: the un-edged graph gives the same bits under the delay -- the test does not reach the hazard"


def test_steps_replay_follows_the_plan():
    """``engine.steps_replay(n)`` takes the plan's steps per graph (engine/plan.py: 20 at C2's and C4's shape, neither
    joined inside a graph) and leaves the engine n steps further either way."""
    m, e = _bench("c2")
    assert e.plan.steps_per_graph == 20 and e.plan.pipe_no_join and e.plan.pipe_prologue == "head"
    e.steps_replay(45)
    torch.cuda.synchronize()
    assert e._pipe is not None and e._pipe.n == 20 and e.st.device_step() == 45
    m4, e4 = _bench("c4")
    assert e4.plan.steps_per_graph == 20 and e4.plan.pipe_no_join and e4.plan.pipe_prologue == "critic"
    e4.steps_replay(43)
    torch.cuda.synchronize()
    assert e4._pipe is not None and e4._pipe.n == 20 and e4.st.device_step() == 43


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_pipelined_bench_path_matches_oracle(name):
    """The replayed 2-step pipelined graph against the pinned fp64 / fp32 oracle: after the replay both engines' gathered
    batches and consumed noise (step 1 in the first engine's buffers, step 2 in its twin's) are read back and the oracle
    takes the two steps from the model's own initial parameters: statistics of both steps <= 1e-4, ``log_alpha`` / PID
    state after step 2, Adam first moments after step 2 at 5e-2 of scale (the bench-path test's later-step gate: they
    blend two steps' gradients from trajectories that differ by round-off)."""
    import bench
    from osrl_amd.engine.pipeline import PipelinedSteps
    from test_gpu_bench_path import BATCH, _oracle
    from test_gpu_train_step import KINK_FLOOR, _note
    wl = bench.Workload(name, torch.device(DEV), 0, 1, None, n_store=1 << 16, use_graph=True)
    m = wl.model
    o64, o32 = _oracle(wl, np.float64), _oracle(wl, np.float32)
    pipe = PipelinedSteps(wl.eng, steps_per_graph=2)
    pipe.run(2)
    torch.cuda.synchronize()
    st64 = []
    for e in pipe.e:
        batch = [getattr(e, k).detach().cpu().numpy().copy() for k in BATCH]
        noise = {k: v.detach().cpu().numpy().copy() for k, v in e.noise.items()}
        st64.append(o64.train_one_step(*batch, noise))
        o32.train_one_step(*batch, noise)
    assert not np.array_equal(pipe.e[0].obs.cpu().numpy(), pipe.e[1].obs.cpu().numpy()), "two steps, two minibatches"
    for s, want in enumerate(st64):
        got = wl.eng.st.read_stats(s + 1)
        for k, r in want.items():
            assert abs(got[k] - r) <= 1e-4 * max(1.0, abs(r)), f"{name} pipelined step {s + 1} {k}: gpu {got[k]} vs oracle {r}"
    groups = {"actor": "opt_actor", "critic": "opt_critic", "cost_critic": "opt_cost", "vae": "opt_vae"}
    worst = {}
    for gname, oname in groups.items():
        grp = m.groups[gname]
        for k, mo in getattr(o64, oname).m.items():
            mg = grp._view(grp.m, k).cpu().numpy()
            scale = max(np.abs(mo).max(), 1e-12)
            d = min(np.abs(mg - mo).max(), np.abs(mg - getattr(o32, oname).m[k]).max())
            worst[gname] = max(worst.get(gname, 0.0), d / scale)
            assert d <= max(5e-2 * scale, 2 * KINK_FLOOR), f"{name} pipelined, first moment {k} after 2 steps: {d:.3e} vs {scale:.3e}"
    _note(f"pipelined bench path {name}, 2 steps in one graph: first-moment diff / scale " +
          ", ".join(f"{g}={v:.2e}" for g, v in worst.items()))
    if wl.cfg["algo"] == "cpq":
        assert abs(m.log_alpha.item() - o64.log_alpha) < 1e-5
    else:
        assert abs(m.controller.error_old - o64.controller.error_old) < 1e-4
        assert abs(m.controller.error_integral - o64.controller.error_integral) < 1e-4
