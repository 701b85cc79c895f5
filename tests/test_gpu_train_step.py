"""GPU parity of whole train steps: osrl_amd (HIP, through the C ABI) vs (a) the golden vectors
captured from the reference and (b) the numpy oracle on the same seeded inputs and injected noise.
Gates: step-1 stats <= 1e-5, <=10-step stats / parameters <= 1e-4."""
import os

import numpy as np
import pytest
import torch

from cases import BEARL_CASES, CASES, COPTIDICE_CASES, make_batch

ALL_CASES = {**CASES, **BEARL_CASES, **COPTIDICE_CASES}
from gpu_util import build_gpu, gpu_batch, gpu_step
from oracle_util import build_oracle, load_golden, oracle_step

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(ALL_CASES))
def test_train_step_matches_golden_and_oracle(name):
    c = ALL_CASES[name]
    g = load_golden(name)
    keys = [str(k) for k in g["stat_keys"]]
    m, tr, lg = build_gpu(c)
    o = build_oracle(c, np.float32)
    b = gpu_batch(c)
    worst = 0.0
    for s in range(c.steps):
        gpu_step(tr, c, b, s)
        ost = oracle_step(o, c, s)
        ref = dict(zip(keys, g["stats"][s]))
        tol = 1e-5 if s == 0 else 1e-4
        for k in keys:
            got = lg.last(k)
            for nm, r in (("golden", ref[k]), ("oracle", ost[k])):
                d = abs(got - r)
                worst = max(worst, d)
                assert d <= tol * max(1.0, abs(r)), f"{name} step {s} {k}: gpu {got} vs {nm} {r} (diff {d:.3e})"
        if f"s{s + 1}/log_alpha" in g:
            assert abs(m.log_alpha.item() - float(g[f"s{s + 1}/log_alpha"])) < 1e-6
        if f"s{s + 1}/tau" in g:
            assert abs(m.tau.item() - float(g[f"s{s + 1}/tau"])) < 1e-5
            assert abs(m.lmbda.item() - float(g[f"s{s + 1}/lmbda"])) < 1e-5
        if f"s{s + 1}/pid_error_old" in g:
            assert abs(m.controller.error_old - float(g[f"s{s + 1}/pid_error_old"])) < 1e-5
            assert abs(m.controller.error_integral - float(g[f"s{s + 1}/pid_error_integral"])) < 1e-5
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        for k, v in sd.items():
            if f"p{s + 1}/{k}" in g:
                d = np.abs(v - g[f"p{s + 1}/{k}"]).max()
                assert d <= 1e-4, f"{name} step {s + 1} param {k}: max diff {d:.3e}"
            elif f"p{s + 1}/smp/{k}" in g:
                d = np.abs(v.reshape(-1)[::97] - g[f"p{s + 1}/smp/{k}"]).max()
                assert d <= 1e-4, f"{name} step {s + 1} param sample {k}: max diff {d:.3e}"
    # final parameters also against the oracle (same-step trajectories)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in o.p.items():
        d = np.abs(sd[k] - v).max()
        assert d <= 1e-4, f"{name} final param {k} vs oracle: {d:.3e}"
    # act()
    bb = make_batch(c)
    with torch.no_grad():
        if c.algo == "bc":
            a = m.actor(b["observations"]).cpu().numpy()
        elif c.algo in ("cpq", "bearl"):
            from osrl_amd import ops
            a = ops.cpq_act(m, b["observations"], True)[0].cpu().numpy()
        elif c.algo == "coptidice":
            a = m.actor(b["observations"], True, True)[0].cpu().numpy()
        else:
            z = torch.from_numpy(g["act_z"]).to(b["observations"].device).clamp(-0.5, 0.5)
            a = m.actor(b["observations"], m.vae.decode(b["observations"], z)).cpu().numpy()
    assert np.abs(a - g["act"]).max() <= 1e-4
    print(f"{name}: worst stat diff {worst:.3e}")


GROUP_GATE = 2e-5  # Adam first moments (= gradients), relative to each tensor's scale
# C2's actor gradient at initialisation is tiny (|m| ~ 2e-6 .. 4e-6 at both seeds tried; the critic head's at one seed
# 4e-4) while ONE row's contribution to it is ~(1 - beta1)/B * |dq/da| ~ 1e-6: a single ReLU unit of one row that sits
# within an ulp of its kink and falls on the other side than in the fp64 oracle moves the sum by ~1e-8 -- with 2048 rows
# x ~1000 units about one such unit is expected per step.  That is what is observed (9.9e-9 and 3.7e-8; the fp32 numpy
# oracle happens to have none: 1.4e-12), so first moments are gated at 2e-5 of the tensor's scale OR this absolute
# floor; every other group of every case (incl. the actor at the C4 shape) meets the relative gate with a 30x margin.
KINK_FLOOR = 1e-7


def _note(msg: str) -> None:
    """Observed margins, kept next to the GPU run's other outputs (gpurun_out/ travels back from the GPU box)."""
    print(msg)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_margins.txt"), "a") as f:
            f.write(msg + "\n")


_C2_ACTOR_FLOOR = {}  # case name -> elements of the actor group that needed the kink floor (filled by the test below)

FULL_CASES = {
    # BASELINE.json C2 / C3 shapes at their full batch sizes: no reference golden (the fixtures stay small), the
    # pinned oracle is the checker.  These are the only parity runs that reach the N*B-row launches (32-row tiles,
    # the capped tile-loop kernel), the paired launches and the 8-wave kernels at bench size.
    "cpq_c2_full": dict(algo="cpq", od=76, ad=2, B=2048, hidden=[256, 256], vae_hidden=400, N=10, steps=1, seed=21),
    # a second seed of C2: the loose actor / critic-head gates of cpq_c2_full are claimed to be a property of THAT
    # seed's gradient (a heavily cancelling batch sum), not of the kernels -- this one takes the tight gate everywhere
    "cpq_c2_full_s2": dict(algo="cpq", od=76, ad=2, B=2048, hidden=[256, 256], vae_hidden=400, N=10, steps=1, seed=31),
    # a third seed: on seed 21 about half of the actor group's elements pass only through the absolute
    # kink floor; test_c2_actor_gradient_needs_no_floor_on_two_of_three_seeds requires that to be the exception
    "cpq_c2_full_s3": dict(algo="cpq", od=76, ad=2, B=2048, hidden=[256, 256], vae_hidden=400, N=10, steps=1, seed=41),
    "bcql_c3_full": dict(algo="bcql", od=33, ad=8, B=4096, hidden=[256, 256], vae_hidden=400, N=10, steps=1, seed=22),
    # BASELINE.json C4 per-GPU shape: CPQ on OfflineHalfCheetah dims (17, 6) -> latent 12, VAE inputs 23 / 29 wide,
    # q inputs 23 wide: other paddings / column-block splits than C2 (cpq_configs.py:359 task)
    "cpq_c4_full": dict(algo="cpq", od=17, ad=6, B=2048, hidden=[256, 256], vae_hidden=400, N=10, steps=1, seed=24),
    # BEAR-Lag at its train-config size (bearl_configs.py: batch 512, N = M = 10)
    "bearl_full": dict(algo="bearl", od=33, ad=8, B=512, hidden=[256, 256], vae_hidden=400, N=10, steps=1, seed=23,
                       hp=dict(mmd_sigma=20.0)),
}


@pytest.mark.parametrize("name", list(FULL_CASES))
def test_full_size_train_step_matches_oracle(name):
    from cases import Case
    c = Case(name, episode_len=1000, **FULL_CASES[name])
    m, tr, lg = build_gpu(c)
    o = build_oracle(c, np.float64)  # fp64: the oracle's own round-off must not blur the comparison at this size
    o32 = build_oracle(c, np.float32)  # the same restatement in fp32: what fp32 round-off alone does to each tensor
    b = gpu_batch(c)
    for s in range(c.steps):
        gpu_step(tr, c, b, s)
        ost = oracle_step(o, c, s)
        oracle_step(o32, c, s)
        for k, r in ost.items():
            got = lg.last(k)
            assert abs(got - r) <= 1e-4 * max(1.0, abs(r)), f"{name} step {s} {k}: gpu {got} vs oracle {r}"
    # Gradients: Adam's first moments after these steps are a fixed linear combination of the per-step gradients, so
    # they are compared directly (relative to each tensor's scale).  The parameters themselves are gated loosely:
    # Adam moves every element by ~lr per step whatever the gradient's size, so an element whose gradient is
    # round-off noise around zero (masked rows, dead ReLUs) may land up to ~2*lr*steps apart between two fp32
    # summation orders.
    from cases import hyper
    hp = hyper(c)
    opts = {"actor": o.opt_actor, "critic": o.opt_critic, "cost_critic": o.opt_cost, "vae": o.opt_vae}
    opts32 = {"actor": o32.opt_actor, "critic": o32.opt_critic, "cost_critic": o32.opt_cost, "vae": o32.opt_vae}
    # Per-tensor gate on max|m_gpu - m_oracle64|: 2e-5 of the tensor's scale -- a 1e-3 relative error in any kernel that
    # first runs at this size (32-row tiles, the 80-row kernel, the paired launches, the 8-wave variants) fails -- or
    # the absolute kink floor above.  Measured on MI355X (gpurun_out/parity_margins.txt): every group of every case
    # sits at 3e-7 .. 7e-7 of its scale except C2's actor (both seeds) and the critic head of one seed.
    worst, worst_ratio, n_floor = {}, {}, {}
    for gname, opt in opts.items():
        grp = m.groups[gname]
        for k, mo in opt.m.items():
            mg = grp._view(grp.m, k).cpu().numpy()
            scale = max(np.abs(mo).max(), 1e-12)
            # a ReLU unit / min-routing decision within an ulp of its kink may fall on either side in fp32 and fp64: the
            # GPU has to agree with ONE of the two restatements (seed 31: GPU == fp32 oracle to 1e-12 where both miss
            # the fp64 oracle by 2.9e-7 on actor.mu_layer.weight)
            d32 = np.abs(opts32[gname].m[k].astype(np.float64) - mo).max()
            d = min(np.abs(mg - mo).max(), np.abs(mg - opts32[gname].m[k]).max())
            worst[gname] = max(worst.get(gname, 0.0), d / scale)
            # HOW MANY elements pass only through the absolute kink floor: a kernel bug hiding under the
            # floor would show as a jump of this count (seed-calibrated: a handful of elements of C2's actor group)
            el = np.minimum(np.abs(mg - mo), np.abs(mg - opts32[gname].m[k]))
            n_floor[gname] = n_floor.get(gname, 0) + int((el > GROUP_GATE * scale).sum())
            if d > GROUP_GATE * scale:
                worst_ratio[gname] = max(worst_ratio.get(gname, 0.0), d)
            assert d <= max(GROUP_GATE * scale, KINK_FLOOR), \
                f"{name} Adam first moment {k} ({gname}): max diff {d:.3e} vs scale {scale:.3e}, fp32 oracle misses by {d32:.3e}"
    _note(f"{name} worst first-moment diff / scale per group: " + ", ".join(f"{g}={v:.2e}" for g, v in worst.items()) +
          ("; beyond 2e-5 of scale (absolute error, gated by the kink floor 1e-7): " +
           ", ".join(f"{g}={v:.2e}" for g, v in worst_ratio.items()) if worst_ratio else "") +
          "; elements that needed the kink floor: " + ", ".join(f"{g}={v}" for g, v in n_floor.items()))
    # Recorded in gpurun_out/parity_margins.txt (round 4, C2 seed 21: actor 41600 of 86532 -- its whole gradient is a
    # heavily cancelling batch sum of scale 1e-5, so 2e-5 of scale is 2e-10 and half its elements sit between that and
    # the 1e-7 floor; critic 93; cost critic and VAE 0).  Gated coarsely: a kernel defect large enough to hide under the
    # floor moves the MAJORITY of a group there, or shows in a group that needs no floor at all today.
    if name.startswith("cpq_c2_full"):
        _C2_ACTOR_FLOOR[name] = n_floor.get("actor", 0)
    for gname, cnt in n_floor.items():
        total = sum(int(np.prod(v.shape)) for v in opts[gname].m.values())
        limit = 0.6 * total if gname == "actor" else max(256, total // 200)
        assert cnt <= limit, f"{name}: {cnt} of {total} elements of {gname} needed the kink floor"
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in o.p.items():
        lr = hp.get(k.split(".")[0].replace("cost_critic", "critic") + "_lr", max(x for n, x in hp.items() if n.endswith("_lr")))
        d = np.abs(sd[k] - v).reshape(-1)
        assert d.max() <= 2.5 * lr * c.steps + 1e-6, f"{name} final param {k} vs oracle: max {d.max():.3e}"
        assert np.median(d) <= 2e-6, f"{name} final param {k}: median diff {np.median(d):.3e}"
    if c.algo in ("cpq", "bearl"):
        assert abs(m.log_alpha.item() - o.log_alpha) < 1e-5


def test_c2_actor_gradient_needs_no_floor_on_two_of_three_seeds():
    """C2's actor gradient at initialisation is a heavily cancelling batch sum (scale ~1e-5): on ONE seed half of its
    elements sit between 2e-5 of that scale and the 1e-7 absolute floor.  That must stay the exception -- at least two
    of the three C2 seeds verify the whole actor group at the RELATIVE gate (zero elements through the floor)."""
    seeds = [k for k in FULL_CASES if k.startswith("cpq_c2_full")]
    if any(k not in _C2_ACTOR_FLOOR for k in seeds):
        pytest.skip("needs the three cpq_c2_full* cases of test_full_size_train_step_matches_oracle in this session")
    clean = [k for k in seeds if _C2_ACTOR_FLOOR[k] == 0]
    _note("C2 actor group, elements through the kink floor per seed: " +
          ", ".join(f"{k}={_C2_ACTOR_FLOOR[k]}" for k in seeds))
    assert len(clean) >= 2, _C2_ACTOR_FLOOR


def test_state_dict_roundtrip_and_keys():
    c = CASES["cpq_small"]
    m, tr, lg = build_gpu(c)
    g = load_golden("cpq_small")
    want = {k[len("p1/"):] for k in g.files if k.startswith("p1/")}
    assert set(m.state_dict().keys()) == want
    b = gpu_batch(c)
    gpu_step(tr, c, b, 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2, tr2, _ = build_gpu(c)
    m2.load_state_dict(sd)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k])


@pytest.mark.parametrize("name", ["bc_small", "cpq_small", "bcql_small"])
def test_graph_replay_is_deterministic_and_trains(name):
    """hipGraph path (device Philox noise): two identical runs agree bit-for-bit; losses are finite;
    the step counter advances once per replay."""
    c = CASES[name]
    outs = []
    for rep in range(2):
        m, tr, lg = build_gpu(c, stats_mode="lazy", use_graph=True)
        b = gpu_batch(c)
        for s in range(6):
            gpu_step(tr, c, b, s, with_noise=False)
        torch.cuda.synchronize()
        eng = m._engine
        assert eng.st.device_step() == 6 and eng.st.host_step == 6
        assert eng.graph is not None, "graph path was not taken"
        vals = {k: [float(x) for x in v] for k, v in lg.data.items()}
        for k, v in vals.items():
            assert len(v) == 6 and all(np.isfinite(v)), (k, v)
        outs.append(({k: v.clone() for k, v in m.state_dict().items()}, vals))
    for k in outs[0][0]:
        assert torch.equal(outs[0][0][k], outs[1][0][k]), f"{name}: {k} differs between identical graph runs"
    assert outs[0][1] == outs[1][1]
    # parameters moved
    p0 = {k: torch.from_numpy(v) for k, v in __import__("cases").make_params(c).items()}
    moved = sum(float((outs[0][0][k].cpu() - p0[k]).abs().max()) > 0 for k in p0)
    assert moved == len(p0)


def test_lazy_statistics_over_several_ring_lengths_equal_the_synchronous_ones():
    """300 graph-replayed steps (ring = 256 rows): the values the lazy logger materialises in bulk flushes are, step
    for step, the ones a synchronising ``stats_mode="sync"`` run logs."""
    c = CASES["cpq_small"]
    logs = {}
    for mode in ("sync", "lazy"):
        m, tr, lg = build_gpu(c, stats_mode=mode, use_graph=True)
        lg.max_keep = 10 ** 6
        b = gpu_batch(c)
        for s in range(300):
            gpu_step(tr, c, b, s, with_noise=False)
        torch.cuda.synchronize()
        logs[mode] = {k: [float(x) for x in v] for k, v in lg.data.items()}
    assert set(logs["sync"]) == set(logs["lazy"])
    for k in logs["sync"]:
        assert len(logs["lazy"][k]) == 300 and logs["lazy"][k] == logs["sync"][k], k


@pytest.mark.parametrize("name", ["cpq_small", "cpq_wide", "bcql_small", "bc_small", "cpq_c2_full", "cpq_c4_full", "bearl_small",
                                  "bearl_wide", "coptidice_small", "coptidice_wide"])
def test_graph_with_parallel_branches_equals_eager_sequential(name):
    """The captured graph (forked side-stream branches, device Philox noise) must produce exactly the same
    parameters as the plain in-order launch sequence: same kernels, same inputs, no atomics."""
    if name in ALL_CASES:
        c = ALL_CASES[name]
    else:  # bench-size CPQ: capped N*B launch beside the VAE phase, paired launches
        from cases import Case
        c = Case(name, episode_len=1000, **FULL_CASES[name])
    res = []
    for use_graph in (False, True):
        m, tr, lg = build_gpu(c, stats_mode="none", use_graph=use_graph)
        b = gpu_batch(c)
        for s in range(4):
            gpu_step(tr, c, b, s, with_noise=False)
        torch.cuda.synchronize()
        assert (m._engine.graph is not None) == use_graph
        if use_graph:
            # the captured launches read their fused-MLP descriptors from HBM (core.ArgArena): every descriptor the
            # capture pass looked up was one the warm-up pass had recorded -- so this comparison is also "kernels
            # reading device-resident argument blocks == kernels taking them by value", bit for bit
            ar = m._engine._arena
            assert ar.ENABLED and ar.blocks > 0 and ar.hits >= ar.blocks and ar.misses == 0, \
                (ar.blocks, ar.hits, ar.misses)
        res.append({k: v.clone() for k, v in m.state_dict().items()})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), f"{name}: {k} differs between eager and graph execution"


def test_data_parallel_path_world1_nccl_matches_single(nccl_world1):
    """The DP wiring (slab pre-reduction -> RCCL all-reduce -> Adam on the reduced gradient, all-gather
    quantile, scalar all-reduces) run as a 1-rank NCCL job must reproduce the plain single-GPU step
    bit-for-bit (all reductions are identities at world_size 1)."""
    import os
    import torch.distributed as dist
    from osrl_amd.engine.dist import DataParallel
    c = CASES["cpq_small"]
    assert nccl_world1.is_initialized()  # the session's one 1-rank NCCL group (conftest.py)
    try:
        res = []
        for use_dp in (False, True):
            m, tr, lg = build_gpu(c)
            b = gpu_batch(c)
            if use_dp:
                dp = DataParallel()
                eng = m.engine(c.B, rows_global=c.B * dp.world, dist=dp)
                dp.broadcast_model(m)
            for s in range(3):
                gpu_step(tr, c, b, s)
            torch.cuda.synchronize()
            res.append(({k: v.clone() for k, v in m.state_dict().items()}, dict(lg.data)))
        for k in res[0][0]:
            assert torch.equal(res[0][0][k], res[1][0][k]), k
        for k in res[0][1]:
            assert [float(x) for x in res[0][1][k]] == [float(x) for x in res[1][1][k]], k
    finally:
        pass


def test_data_parallel_graph_capture_world1(nccl_world1):
    """hipGraph capture of the DP step including its RCCL collectives (1-rank job): must either capture
    and replay deterministically or fall back to eager with a warning -- never hang or corrupt state."""
    import os
    import torch.distributed as dist
    from osrl_amd.engine.dist import DataParallel
    c = CASES["cpq_small"]
    assert nccl_world1.is_initialized()  # the session's one 1-rank NCCL group (conftest.py)
    try:
        outs = []
        for rep in range(2):
            m, tr, lg = build_gpu(c, stats_mode="none", use_graph=True)
            dp = DataParallel()
            eng = m.engine(c.B, rows_global=c.B, dist=dp)
            b = gpu_batch(c)
            for s in range(4):
                gpu_step(tr, c, b, s, with_noise=False)
            torch.cuda.synchronize()
            assert eng.st.device_step() == 4
            outs.append({k: v.clone() for k, v in m.state_dict().items()})
            print("DP graph captured:", eng.graph is not None)
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), k
    finally:
        pass


@pytest.mark.parametrize("name", ["bcql_pid", "bc_small", "bearl_lap"])
def test_data_parallel_world1_other_algos(name, nccl_world1):
    """BCQ-Lag / BEAR-Lag (global-mean pre-pass for the PID controller and the dual step) and BC under the DP hook as
    a 1-rank NCCL job == single GPU."""
    import os
    import torch.distributed as dist
    from osrl_amd.engine.dist import DataParallel
    c = ALL_CASES[name]
    assert nccl_world1.is_initialized()  # the session's one 1-rank NCCL group (conftest.py)
    try:
        res = []
        for use_dp in (False, True):
            m, tr, lg = build_gpu(c)
            b = gpu_batch(c)
            if use_dp:
                m.engine(c.B, rows_global=c.B, dist=DataParallel())
            for s in range(3):
                gpu_step(tr, c, b, s)
            torch.cuda.synchronize()
            res.append(({k: v.clone() for k, v in m.state_dict().items()}, {k: [float(x) for x in v] for k, v in lg.data.items()}))
        for k in res[0][0]:
            assert torch.equal(res[0][0][k], res[1][0][k]), k
        for k in res[0][1]:
            assert np.allclose(res[0][1][k], res[1][1][k], rtol=1e-6, atol=1e-7), k
    finally:
        pass


# --------------------------------------------------------------------------- #
# checkpoint round trip + resume (osrl_amd/common/checkpoint.py)
# --------------------------------------------------------------------------- #
def _train_state(m):
    out = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for name, g in m.groups.items():
        out["m/" + name], out["v/" + name] = g.m.clone(), g.v.clone()
    for k in ("log_alpha", "pid_state", "scalar_leaves"):
        if isinstance(getattr(m, k, None), torch.Tensor):
            out[k] = getattr(m, k).clone()
    return out


@pytest.mark.parametrize("name", ["bc_small", "cpq_small", "bcql_pid", "bearl_lap", "coptidice_small"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_checkpoint_resume_is_bit_identical(name, use_graph, tmp_path):
    """3 steps -> save -> load into a fresh model -> 2 steps  ==  5 uninterrupted steps, bit for bit (parameters,
    targets, Adam moments, log_alpha / PID state); the noise is the device Philox stream, keyed by the step count
    the checkpoint carries.  The file keeps the reference's {"model_state": ...} layout."""
    from osrl_amd.common.checkpoint import load_checkpoint, save_checkpoint
    c = ALL_CASES[name]
    b = gpu_batch(c)
    m_a, tr_a, _ = build_gpu(c, use_graph=use_graph)
    for s in range(5):
        gpu_step(tr_a, c, b, s, with_noise=False)
    m_b, tr_b, _ = build_gpu(c, use_graph=use_graph)
    for s in range(3):
        gpu_step(tr_b, c, b, s, with_noise=False)
    path = str(tmp_path / "model.pt")
    save_checkpoint(m_b, path)
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert set(raw) == {"model_state", "osrl_amd"} and raw["osrl_amd"]["step"] == 3
    assert set(raw["model_state"]) == set(m_b.state_dict())
    m_c, tr_c, _ = build_gpu(c, use_graph=use_graph)
    load_checkpoint(m_c, path)
    for s in range(3, 5):
        gpu_step(tr_c, c, b, s, with_noise=False)
    torch.cuda.synchronize()
    sa, sc = _train_state(m_a), _train_state(m_c)
    assert set(sa) == set(sc)
    for k in sa:
        assert torch.equal(sa[k], sc[k]), k
    assert m_c._engine.st.device_step() == 5
    # weights only (what a reference checkpoint holds): loads, the optimizer starts fresh
    m_d, tr_d, _ = build_gpu(c)
    load_checkpoint(m_d, {"model_state": raw["model_state"]})
    for k, v in m_d.state_dict().items():
        assert torch.equal(v.cpu(), raw["model_state"][k]), k
    assert all(float(g.m.abs().max()) == 0.0 for g in m_d.groups.values())


def test_engine_rebuild_at_a_smaller_batch_does_not_read_stale_gradient_slabs():
    """ADVICE r2 (core.py DwPlan.launch): the dW kernels STORE rows [0, own splits) of the group's slab tensor and Adam
    sums ``cur_splits`` rows.  An engine rebuilt at a smaller batch has fewer splits; rows beyond them still held the
    previous engine's partial gradients in the retained (grow-only) tensor.  After the fix the rebuilt engine steps
    exactly like a model that never saw the big batch."""
    from cases import Case
    c = Case("rebuild", "cpq", od=17, ad=6, B=2048, hidden=[64, 64], vae_hidden=96, N=4, steps=1, episode_len=1000)
    small = 256
    outs = []
    for pre_steps in (2, 0):
        m, tr, _ = build_gpu(c, stats_mode="none", use_graph=False)
        b = gpu_batch(c)
        for s in range(pre_steps):  # big batch first: several row splits per group, slabs filled
            gpu_step(tr, c, b, s, with_noise=False)
        if pre_steps:
            assert max(g.n_splits for g in m.groups.values()) > 1
            # bring parameters / moments / step count back to the fresh state, keep the (dirty) slab tensors
            m2, _, _ = build_gpu(c, stats_mode="none", use_graph=False)
            for n, g in m.groups.items():
                g2 = m2.groups[n]
                g.p.copy_(g2.p); g.m.zero_(); g.v.zero_()
                if g.tgt is not None:
                    g.tgt.copy_(g2.tgt)
            m.log_alpha.copy_(m2.log_alpha)
            m.repack()
            m._engine.st.set_step(0)
        bs = {k: v[:small].contiguous() for k, v in b.items()}
        eng = m.engine(small)
        if pre_steps:
            eng.st.set_step(0)
        for s in range(2):
            eng.step(bs["observations"], bs["next_observations"], bs["actions"], bs["rewards"], bs["costs"], bs["done"],
                     use_graph=False)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in m.state_dict().items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{k}: the rebuilt engine read stale slab rows"


def test_engine_rebuild_keeps_the_step_count():
    """A new batch size rebuilds the launch plan; Adam's step count (bias correction) must carry over."""
    c = CASES["bc_small"]
    m, tr, _ = build_gpu(c)
    b = gpu_batch(c)
    for s in range(3):
        tr.train_one_step(b["observations"], b["actions"])
    tr.train_one_step(b["observations"][:16], b["actions"][:16])
    assert m._engine.B == 16 and m._engine.st.device_step() == 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cpq_small", "cpq_odd", "cpq_wide", "cpq_c2_full", "cpq_c4_full", "bcql_small", "bcql_pid",
                                  "bcql_c3_full", "bearl_small", "bearl_lap", "bearl_full"])
def test_seeded_backward_launches_equal_the_loss_launches(name):
    """Round 4: the CPQ step's backward launches compute the gradient they start from (osrl_mlp_backward_dz_seed) instead
    of reading it from five loss launches (vae_loss, cpq_critic_loss, cpq_cost_loss, cpq_actor_loss, gauss_head_bwd).
    Same expressions per row => every parameter, target and Adam moment BIT-equal to the plan with the loss launches
    after each of three steps; the logged statistics are the same sums in another fixed order (<= 2e-6 relative)."""
    from cases import Case
    from osrl_amd.engine import glue as G
    c = Case(name, episode_len=1000, **FULL_CASES[name]) if name in FULL_CASES else ALL_CASES[name]
    b = gpu_batch(c)
    runs = []
    for seeds in (False, True):
        old, old_ns = G.SEEDS, G.VAE_NS_AUTO
        G.SEEDS = seeds
        # (round 5: the all-CU VAE launches exist only in the seeded plan and sum in another order -- this comparison is
        # about the seeds, so both runs take the four fused VAE launches; test_vae_ns_launches_equal_the_fused_launches
        # and the OSRL_VAE_NS=1 runs of the full-size oracle tests cover the other form)
        G.VAE_NS_AUTO = False
        try:
            m, tr, lg = build_gpu(c)
            stats = []
            for s in range(3):
                gpu_step(tr, c, b, s)
                stats.append({k: float(lg.last(k)) for k in m._engine.st.keys})
            sd = m._engine.seeds if hasattr(m._engine, "seeds") else m._engine._seeds()
            assert (sd is not None) == seeds
            torch.cuda.synchronize()
            runs.append((m, stats))
        finally:
            G.SEEDS, G.VAE_NS_AUTO = old, old_ns
    (ma, sa), (mb, sb) = runs
    for gname in ma.groups:
        ga, gb = ma.groups[gname], mb.groups[gname]
        for buf in ("p", "m", "v", "tgt"):
            x, y = getattr(ga, buf), getattr(gb, buf)
            if x is not None:
                assert torch.equal(x, y), (gname, buf, float((x - y).abs().max()))
    for scalar in ("log_alpha", "pid_state"):  # the dual variables / PID integrators see the same gradients' statistics
        if isinstance(getattr(ma, scalar, None), torch.Tensor):
            assert torch.equal(getattr(ma, scalar), getattr(mb, scalar)), scalar
    for s, (x, y) in enumerate(zip(sa, sb)):
        for k in x:
            assert abs(x[k] - y[k]) <= 2e-6 * max(1.0, abs(x[k])), (s, k, x[k], y[k])


def test_an_engine_superseded_on_its_model_refuses_to_step():
    """ADVICE r3 (core.py ensure_slabs): once another engine's dW plans are attached to a model's gradient slabs, an
    engine built earlier -- its plans' split counts, its captured graph -- must not run any more; it raises instead of
    summing slab rows the newcomer writes."""
    c = ALL_CASES["cpq_small"]
    m, tr, lg = build_gpu(c)
    b = gpu_batch(c)
    gpu_step(tr, c, b, 0)
    old = m._engine
    half = {k: v[: c.B // 2].contiguous() for k, v in b.items()}
    tr.train_one_step(half["observations"], half["next_observations"], half["actions"], half["rewards"], half["costs"],
                      half["done"])  # another batch size -> model.engine() builds a new engine on the same groups
    assert m._engine is not old
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="stale"):
        old.step(b["observations"], b["next_observations"], b["actions"], b["rewards"], b["costs"], b["done"])


def _random_cases():
    """Seeded random (obs_dim, act_dim, batch, widths, N) tuples -- north_star says tuples, plural -- plus three picked to
    land where the plan chooser (engine/plan.py) switches forms: inside the all-CU VAE region with a ragged last row tile,
    at its row-count edges, and on a narrow VAE with one 80-column group."""
    from cases import Case
    rs = np.random.RandomState(505)
    hid = [[256, 256], [64, 64], [48, 32], [128, 128], [256, 256], [96, 160]]
    vae = [400, 80, 48, 160, 240, 400, 320]
    out = []
    for i in range(12):
        algo = "bcql" if i % 3 == 2 else "cpq"
        out.append(Case(f"rand{i}_{algo}", algo, od=int(rs.randint(3, 90)), ad=int(rs.randint(1, 9)),
                        B=int(rs.choice([24, 100, 256, 512, 1024, 1200, 2048])), hidden=hid[rs.randint(len(hid))],
                        vae_hidden=int(vae[rs.randint(len(vae))]), N=int(rs.choice([3, 10])), steps=1, episode_len=1000,
                        seed=100 + i))
    out += [Case("rand_ns_ragged_cpq", "cpq", od=40, ad=3, B=1200, hidden=[128, 128], vae_hidden=240, N=3, steps=1,
                 episode_len=1000, seed=120),
            Case("rand_ns_edge_cpq", "cpq", od=30, ad=8, B=1024, hidden=[64, 64], vae_hidden=160, N=10, steps=1,
                 episode_len=1000, seed=121),
            Case("rand_ns_one_group_bcql", "bcql", od=11, ad=5, B=2048, hidden=[48, 32], vae_hidden=80, N=3, steps=1,
                 episode_len=200, seed=122)]
    return out


@pytest.mark.parametrize("c", _random_cases(), ids=lambda c: f"{c.name}-{c.od}x{c.ad}-B{c.B}-V{c.vae_hidden}")
def test_random_shape_tuples_match_the_oracle(c, monkeypatch):
    """One train step of whatever plan the chooser picks for an arbitrary shape == the pinned oracle (fp32 and fp64, the
    closer one): statistics <= 1e-5; the GRADIENTS (Adam first moments after this one step) at ``GROUP_GATE`` = 2e-5 of
    each tensor's scale or the absolute kink floor, as in the full-size tests; the dual variable / PID state.  The
    parameters themselves are only bounded by what Adam can do to an element whose gradient is round-off noise (it moves
    by ~lr whatever the gradient's size: max 2.5 lr apart, median <= 2e-6) -- the 1e-4 claim of north_star is carried by
    the statistics and the gradients, not by that bound.  With the pinned rows of engine/plan.py
    (tests/test_host_cpu.py) this leaves no reachable plan untested."""
    if c.name.startswith("rand_ns"):  # the all-CU VAE launches on request: BCQ-Lag always, CPQ away from the one hidden
        monkeypatch.setenv("OSRL_LAB", "1")  # width (400) the chooser's rule was timed at (engine/plan.py vae_ns_auto)
        monkeypatch.setenv("OSRL_VAE_NS", "1")
    m, tr, lg = build_gpu(c)
    o32, o64 = build_oracle(c, np.float32), build_oracle(c, np.float64)
    b = gpu_batch(c)
    gpu_step(tr, c, b, 0)
    from oracle.osrl_oracle import MLP, KinkBook
    book = KinkBook(ulps=2.0)  # which ReLU units sit within fp32 round-off of their kink, and what they may change
    s32 = oracle_step(o32, c, 0)
    MLP.kink = book
    try:
        s64 = oracle_step(o64, c, 0)
    finally:
        MLP.kink = None
    eng = m._engine
    if c.name.startswith("rand_ns"):
        assert eng.vae_ns is not None, "this case is meant to run the all-CU VAE launches"
    _note(f"{c.name}: plan {eng.plan}; all-CU VAE launches {'on' if eng.vae_ns is not None else 'off'}")
    if eng.plan.vae_ns:
        assert eng.vae_ns is not None, "the plan chose the all-CU VAE launches but the library refused the shape"
    for k in s64:
        got = lg.last(k)
        d = min(abs(got - s64[k]), abs(got - s32[k]))
        assert d <= 1e-5 * max(1.0, abs(s64[k])), f"{c.name} {k}: gpu {got} vs oracle {s64[k]} / {s32[k]}"
    opt_names = {"actor": "opt_actor", "critic": "opt_critic", "cost_critic": "opt_cost", "vae": "opt_vae"}
    # A miss of the strict gate is admitted only where the oracle itself says a ReLU unit within 2 ulp of its kink can move
    # THAT element, and only by that much: KinkBook.allow bounds, per gradient element, what flipping the undecidable
    # units changes (row j of dW by |dy[r, j]| |x[r, :]|, carried down the MLP) -- no blanket budget.
    worst, n_floor, n_kink = {}, 0, 0
    n_near = sum(len(r) for calls in book.near.values() for r, _ in calls)
    for gname, oname in opt_names.items():
        grp = m.groups[gname]
        for k, mo in getattr(o64, oname).m.items():
            mg = grp._view(grp.m, k).cpu().numpy()
            scale = max(np.abs(mo).max(), 1e-12)
            el = np.minimum(np.abs(mg - mo), np.abs(mg - getattr(o32, oname).m[k]))
            worst[gname] = max(worst.get(gname, 0.0), el.max() / scale)
            strict = max(GROUP_GATE * scale, KINK_FLOOR)
            n_floor += int((el > GROUP_GATE * scale).sum())
            n_kink += int((el > strict).sum())
            allow = (1.0 - 0.9) * np.asarray(book.allow.get(k, 0.0)) * 1.01
            over = el - (strict + allow)
            assert over.max() <= 0, \
                f"{c.name} Adam first moment {k} ({gname}): max diff {el.max():.3e} vs scale {scale:.3e}; " \
                f"{int((over > 0).sum())} element(s) beyond the strict gate + what the {n_near} near-kink unit(s) can move"
    _note(f"{c.name}: first-moment diff / scale " + ", ".join(f"{g}={v:.2e}" for g, v in worst.items()) +
          f"; elements that needed the kink floor: {n_floor}; beyond it, covered by the {n_near} near-kink unit(s)' bound: {n_kink}")
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in o64.p.items():
        d = min(np.abs(sd[k] - v).max(), np.abs(sd[k] - o32.p[k]).max())
        # (Adam moves an element whose gradient is round-off noise by up to lr either way: 2 lr apart at most)
        assert d <= 2.5e-3 and np.median(np.abs(sd[k] - v)) <= 2e-6, f"{c.name} param {k}: {d:.3e}"
    if c.algo == "cpq":
        assert abs(m.log_alpha.item() - o64.log_alpha) < 1e-5
