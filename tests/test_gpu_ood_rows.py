"""CPQ's OOD penalty on the rows that count (engine/plan.py ``ood_rows``: the rule on one GPU wherever the pipelined graphs are
not joined -- C2, C4; OSRL_LAB=1 OSRL_OOD_ROWS=0 / 1 forces it).

``qc_ood = ((KL_loss >= quantile) * qc_sampled).mean(0)`` (cpq.py:183-184, under no_grad) multiplies three quarters of the
N*B target-cost-critic outputs by zero.  The plan runs the VAE encoder on the N*B rows first, takes the quantile and the
list of rows that reach it in one launch (osrl_cpq_ood_select), runs ``cost_critic_old`` on that list only
(osrl_rows_t.row_list) and sums its outputs; the cost critics' Polyak step moves behind this last reader of the old targets
(osrl_polyak).  The term has no gradient to any network (cpq.py:186-187): every parameter, moment and target of the step must
come out BIT-EQUAL to the default plan's, only ``log_alpha`` and the logged cost-critic loss may differ by the order of one
sum.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _wl(name, monkeypatch, on):
    import bench
    monkeypatch.setenv("OSRL_LAB", "1")  # (the rule turns the plan on where the side branch is the long one: C4)
    monkeypatch.setenv("OSRL_OOD_ROWS", "1" if on else "0")
    wl = bench.Workload(name, torch.device(DEV), 0, 1, None, n_store=1 << 16, use_graph=True)
    assert bool(wl.eng.ood_rows) == on and bool(wl.eng.plan.ood_rows) == on
    return wl


def _state(m):
    out = {}
    for n, g in m.groups.items():
        out[n + ".p"], out[n + ".m"], out[n + ".v"] = g.p.clone(), g.m.clone(), g.v.clone()
        if g.tgt is not None:
            out[n + ".tgt"], out[n + ".tf"] = g.tgt.clone(), g.tf.clone()
    return out


@pytest.mark.parametrize("name", ["c2", "c4"])
def test_ood_rows_plan_leaves_every_network_bit_equal(name, monkeypatch):
    steps = 4
    res = []
    for on in (False, True):
        wl = _wl(name, monkeypatch, on)
        for _ in range(steps):
            wl.eng.step_replay(True)
        torch.cuda.synchronize()
        assert wl.eng.graph is not None
        stats = [wl.eng.st.read_stats(s) for s in range(1, steps + 1)]
        cnt = int(wl.eng.ood_count[0].item())
        res.append((_state(wl.model), stats, float(wl.model.log_alpha.item()), cnt))
        del wl
        torch.cuda.empty_cache()
    (s0, st0, a0, _), (s1, st1, a1, cnt) = res
    n = 10 * 2048
    assert n - int(0.75 * (n - 1)) - 1 <= cnt <= n - int(0.75 * (n - 1)) + 8, cnt  # a quarter of the rows (+ ties)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), f"{name}: {k} differs between the plans ({(s0[k] - s1[k]).abs().max().item():.3e})"
    assert abs(a0 - a1) <= 1e-6, (a0, a1)
    for a, b in zip(st0, st1):
        for k in a:
            tol = 1e-5 * max(1.0, abs(a[k])) if k in ("loss/cost_critic_loss", "loss/alpha_value") else 0.0
            assert abs(a[k] - b[k]) <= tol, (k, a[k], b[k])


def test_ood_rows_plan_matches_the_oracle_and_pipelines(monkeypatch):
    """The replayed step of the opt-in plan against the pinned oracle on its read-back batch (as
    tests/test_gpu_bench_path.py does for the default plan), and several steps per graph == single-step replays."""
    from test_gpu_bench_path import BATCH, _oracle
    wl = _wl("c2", monkeypatch, True)
    o64 = _oracle(wl, np.float64)
    wl.eng.step_replay(True)
    torch.cuda.synchronize()
    batch = [getattr(wl.eng, k).detach().cpu().numpy().copy() for k in BATCH]
    noise = {k: v.detach().cpu().numpy().copy() for k, v in wl.eng.noise.items()}
    want = o64.train_one_step(*batch, noise)
    got = wl.eng.st.read_stats(1)
    for k, r in want.items():
        assert abs(got[k] - r) <= 1e-4 * max(1.0, abs(r)), f"{k}: gpu {got[k]} vs oracle {r}"
    assert abs(wl.model.log_alpha.item() - o64.log_alpha) < 1e-5
    del wl
    torch.cuda.empty_cache()
    a, b = _wl("c2", monkeypatch, True), _wl("c2", monkeypatch, True)
    for _ in range(4):
        a.eng.step_replay(True)
    b.eng.steps_replay(4, steps_per_graph=2)
    torch.cuda.synchronize()
    sa, sb = _state(a.model), _state(b.model)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert a.model.log_alpha.item() == b.model.log_alpha.item()
