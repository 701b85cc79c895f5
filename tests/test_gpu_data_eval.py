"""GPU tests of the minibatch sources (on-device CDT window sampler vs the numpy restatement of
SequenceDataset.__prepare_sample) and of evaluate()/rollout()/act() against the oracle policy on the
build-owned synthetic environment."""
import numpy as np
import pytest
import torch

from cases import BEARL_CASES, CASES, CDT_CASES, COPTIDICE_CASES, make_cdt_params
from gpu_util import build_gpu
from oracle.osrl_oracle import prepare_sequence_sample, rollout
from oracle_util import build_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_sequence_store_matches_prepare_sample():
    from osrl_amd.common.replay import SequenceStore
    from osrl_amd.engine.core import StepState
    rs = np.random.RandomState(0)
    od, ad, T, B = 5, 3, 8, 512
    trajs = []
    for i in range(37):
        L_ = int(rs.randint(1, 30))
        trajs.append(dict(observations=rs.randn(L_, od).astype(np.float32), actions=rs.randn(L_, ad).astype(np.float32),
                          returns=rs.rand(L_).astype(np.float32) * 100, cost_returns=rs.rand(L_).astype(np.float32) * 20,
                          costs=(rs.rand(L_) < 0.3).astype(np.float32)))
    prob = rs.rand(37)
    prob /= prob.sum()
    for sp in (None, prob):
        store = SequenceStore(trajs, T, DEV, reward_scale=0.1, cost_scale=2.0, sample_prob=sp, seed=3)
        st = StepState(DEV, ["x"])
        st.tick()
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
        outs = (z(B, T, od), z(B, T, ad), z(B, T), z(B, T), z(B, T, dt=torch.int64), z(B, T), z(B), z(B, T))
        idx = z(B, 2, dt=torch.int32)
        store.gather(*outs, st.ptr, idx_out=idx)
        torch.cuda.synchronize()
        ii = idx.cpu().numpy()
        o = [x.cpu().numpy() for x in outs]
        for b in range(B):
            tr, start = int(ii[b, 0]), int(ii[b, 1])
            assert 0 <= tr < 37 and 0 <= start < len(trajs[tr]["costs"])
            ref = prepare_sequence_sample(trajs[tr], start, T, 0.1, 2.0)
            for got, want in zip(o, ref):
                np.testing.assert_allclose(got[b], np.asarray(want, np.float64), rtol=1e-6, atol=1e-6)
        counts = np.bincount(ii[:, 0], minlength=37) / B
        target = prob if sp is not None else np.full(37, 1 / 37)
        assert np.abs(counts - target).max() < 0.06


@pytest.mark.parametrize("name", ["bc_small", "cpq_small"])
def test_evaluate_matches_oracle_policy(name):
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv
    c = CASES[name]
    m, tr, lg = build_gpu(c)
    o = build_oracle(c)
    env_g, env_o = SyntheticSafeEnv(c.od, c.ad, 40, seed=1), SyntheticSafeEnv(c.od, c.ad, 40, seed=1)
    tr.env = env_g
    m.episode_len = 40
    ret, cost, ln = tr.evaluate(2)
    pol = (lambda ob: o.act(ob[None])[0])
    r0, n0, c0 = rollout(pol, env_o, 40)
    if name != "bc_small":
        r0, c0 = r0 / tr.reward_scale, c0 / tr.cost_scale
    assert ln == n0 and abs(ret - r0) < 1e-3 * max(1, abs(r0)) and abs(cost - c0) < 0.5, (ret, r0, cost, c0)


@pytest.mark.parametrize("case", ["cdt_small", "cdt_v_prefix_det", "cdt_v_norew"])
def test_cdt_rollout_matches_oracle(case):
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv
    from test_gpu_cdt import build_cdt_gpu
    from test_oracle_cdt_golden import build_cdt_oracle
    c = CDT_CASES[case]
    m, tr, lg = build_cdt_gpu(c)
    o = build_cdt_oracle(c)
    EL = 12
    m.episode_len = EL
    env_g, env_o = SyntheticSafeEnv(c.od, c.ad, EL, seed=2), SyntheticSafeEnv(c.od, c.ad, EL, seed=2)
    tr.env = env_g
    ret, cost, ln = tr.evaluate(1, target_return=30.0, target_cost=5.0)

    # the same rollout (cdt.py:436-518) driven by the numpy oracle
    T = c.T
    S, A = np.zeros((EL + 1, c.od), np.float32), np.zeros((EL, c.ad), np.float32)
    R, C = np.zeros(EL + 1, np.float32), np.zeros(EL + 1, np.float32)
    obs, _ = env_o.reset()
    S[0], R[0], C[0] = obs, 30.0, 5.0
    r0 = c0 = 0.0
    for step in range(EL):
        lo = max(0, step + 1 - T)
        n = step + 1 - lo
        pad = lambda x: np.concatenate([x, np.zeros((T - n,) + x.shape[1:], x.dtype)])[None]  # noqa: E731
        mask = pad(np.ones(n, np.float32))
        acts = o.act_mean(pad(S[lo:step + 1]), pad(A[lo:step + 1]), pad(R[lo:step + 1]), pad(C[lo:step + 1]),
                          pad(np.arange(lo, step + 1)), mask, np.array([5.0], np.float32))
        act = np.clip(acts[0, n - 1], -1, 1)
        obs, reward, term, trunc, info = env_o.step(act)
        A[step], S[step + 1] = act, obs
        R[step + 1], C[step + 1] = R[step] - reward, C[step] - info["cost"]
        r0 += reward
        c0 += info["cost"]
    assert ln == EL and abs(ret - r0 / tr.reward_scale) < 1e-3 * max(1, abs(r0)) and abs(cost - c0) < 0.5


# --------------------------------------------------------------------------- #
# batched on-device evaluate()
# --------------------------------------------------------------------------- #
def test_env_step_kernel_matches_numpy_env():
    """csrc/env.hip vs the scalar numpy environment, episode by episode (clip, latch at episode_len, totals)."""
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    rs = np.random.RandomState(0)
    for od, ad, E, EL in ((5, 2, 7, 9), (76, 2, 33, 6), (130, 8, 4, 5)):
        env = SyntheticSafeEnv(od, ad, EL, seed=3, init_noise=0.5)
        venv = VecSyntheticSafeEnv(env, E, DEV, base_seed=10)
        obs = torch.full((E, od + 1), 7.0, device=DEV)  # one extra column that must stay untouched
        venv.reset(obs)
        d = venv.desc(EL, 2.0)
        scal = [SyntheticSafeEnv(od, ad, EL, seed=3, init_noise=0.5) for _ in range(E)]
        tot = np.zeros((E, 3))
        for e, sc in enumerate(scal):
            o0, _ = sc.reset(seed=10 + e)
            np.testing.assert_array_equal(obs[e, :od].cpu().numpy(), o0)
        for step in range(EL + 2):  # two steps past the end: finished episodes are frozen
            a = (rs.randn(E, ad) * 1.5).astype(np.float32)
            venv.step(d, torch.tensor(a, device=DEV), obs)
            for e, sc in enumerate(scal):
                if step < EL:
                    o, r, term, trunc, info = sc.step(a[e])
                    tot[e] += (r, info["cost"] * 2.0, 1)
                    assert trunc == (step == EL - 1)
            got = obs.cpu().numpy()
            want = np.stack([sc.s for sc in scal])
            np.testing.assert_allclose(got[:, :od], want, rtol=1e-5, atol=1e-5)
            assert (got[:, od] == 7.0).all()
        acc = venv.acc.cpu().numpy()
        np.testing.assert_allclose(acc[:, 0], tot[:, 0], rtol=1e-5, atol=1e-4)
        np.testing.assert_array_equal(acc[:, 1:3], tot[:, 1:3])
        assert (acc[:, 3] == 1.0).all()


def _oracle_episode(policy, env, seed, episode_len, cost_scale=1.0):
    obs, _ = env.reset(seed=seed)
    ret = cost = 0.0
    n = 0
    for _ in range(episode_len):
        obs, r, term, trunc, info = env.step(policy(obs))
        ret += r
        cost += info["cost"] * cost_scale
        n += 1
        if term or trunc:
            break
    return ret, cost, n


@pytest.mark.parametrize("name", ["bc_small", "cpq_small", "bcql_small", "bearl_lap", "coptidice_small"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_batched_evaluate_matches_oracle_rollouts(name, use_graph):
    """trainer.evaluate(E) on a VecSyntheticSafeEnv == the oracle policy rolled out episode by episode on the
    scalar environment from the same E initial states."""
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    from osrl_amd.engine.rollout import BatchedRollout
    c = {**CASES, **BEARL_CASES, **COPTIDICE_CASES}[name]
    m, tr, lg = build_gpu(c, use_graph=use_graph)
    o = build_oracle(c)
    E, EL = 24, 30
    m.episode_len = EL
    env = SyntheticSafeEnv(c.od, c.ad, 50, seed=1, init_noise=0.7)
    venv = VecSyntheticSafeEnv(env, E, DEV, base_seed=100)
    tr.env = venv
    cs = 1.0
    if name == "bcql_small":
        rs = np.random.RandomState(5)
        z = rs.randn(E, m.latent_dim).astype(np.float32)  # one fixed decode noise per episode (incl. |z| > 0.5)
        ro = BatchedRollout(m, venv, "bcql", cs, z=torch.tensor(z, device=DEV), use_graph=use_graph)
        rets, costs, lens = ro.run()
        pols = [(lambda ob, e=e: o.act(ob[None], z[e][None])[0]) for e in range(E)]
    else:
        ret, cost, ln = tr.evaluate(E)
        rets, costs, lens = tr._rollout[1].run()  # second run of the cached rollout: reset + replay
        assert abs(ret - rets.mean()) < 1e-6 and cost == costs.mean() and ln == lens.mean() == EL
        pols = [(lambda ob: o.act(ob[None])[0])] * E
    ref = np.array([_oracle_episode(pols[e], SyntheticSafeEnv(c.od, c.ad, 50, seed=1, init_noise=0.7), 100 + e, EL, cs)
                    for e in range(E)])
    np.testing.assert_array_equal(lens, ref[:, 2])
    np.testing.assert_allclose(rets, ref[:, 0], rtol=1e-4, atol=1e-3)
    assert np.abs(costs - ref[:, 1]).max() <= 1.0 and (costs != ref[:, 1]).mean() <= 0.1  # indicator at a threshold
    assert np.unique(np.round(rets, 3)).size > E // 2, "episodes must differ (distinct initial states)"
    with pytest.raises(ValueError):
        tr.evaluate(E + 1)


def test_batched_evaluate_bc_multitask_and_bcql_noise():
    """BC multi-task appends cost_limit to every observation (bc.py:132-138); BCQL without a fixed z draws fresh
    device noise per env step (two runs with the same seed agree, the episodes of one run do not)."""
    from osrl_amd.algorithms import BC, BCTrainer
    from osrl_amd.common.logger import DummyLogger
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    from osrl_amd.engine.rollout import BatchedRollout
    od, ad, E, EL = 6, 2, 8, 12
    torch.manual_seed(0)
    m = BC(od + 1, ad, 1.0, [32, 32], EL, device=DEV)
    tr = BCTrainer(m, None, DummyLogger(), actor_lr=1e-3, bc_mode="multi-task", cost_limit=20, device=DEV)
    env = SyntheticSafeEnv(od, ad, EL, seed=4, init_noise=0.5)
    tr.env = VecSyntheticSafeEnv(env, E, DEV, base_seed=0)
    got = tr.evaluate(E)
    tr2 = BCTrainer(m, SyntheticSafeEnv(od, ad, EL, seed=4, init_noise=0.5), DummyLogger(), actor_lr=1e-3,
                    bc_mode="multi-task", cost_limit=20, device=DEV)
    rets = []
    for e in range(E):  # the reference-shaped scalar loop, same initial states
        obs, _ = tr2.env.reset(seed=e)
        ret = 0.0
        for _ in range(EL):
            obs, r, *_ = tr2.env.step(m.act(np.append(obs, 20.0).astype(np.float32)))
            ret += r
        rets.append(ret)
    assert abs(got[0] - np.mean(rets)) < 1e-3 * max(1.0, abs(np.mean(rets))) and got[2] == EL

    c = CASES["bcql_small"]
    mb, trb, _ = build_gpu(c)
    mb.episode_len = EL
    venv = VecSyntheticSafeEnv(SyntheticSafeEnv(c.od, c.ad, EL, seed=2), E, DEV)  # identical initial states
    ro = BatchedRollout(mb, venv, "bcql", seed=9)
    r1 = ro.run()[0]
    ro2 = BatchedRollout(mb, venv, "bcql", seed=9)
    r2 = ro2.run()[0]
    np.testing.assert_array_equal(r1, r2)
    assert np.unique(r1).size > 1, "per-episode decode noise must differ"


# --------------------------------------------------------------------------- #
# dataset ingestion on device vs the reference's outputs (golden) and the numpy oracle
# --------------------------------------------------------------------------- #
def _np(t):
    return t.cpu().numpy()


def test_ingest_sequence_dataset_matches_reference_golden():
    from cases import make_ingest_dataset
    from oracle_util import load_golden
    from osrl_amd.common.ingest import compute_cost_sample_prob, process_sequence_dataset
    g = load_golden("ingest")
    for rev, tag in ((False, "fwd"), (True, "rev")):
        tb = process_sequence_dataset(make_ingest_dataset(), rev, DEV)
        lens = g[f"seq_{tag}_len"]
        assert np.array_equal(_np(tb["traj_len"]), lens)
        assert np.array_equal(_np(tb["traj_start"]), np.concatenate([[0], np.cumsum(lens)[:-1]]))
        for k in ("observations", "actions", "rewards", "costs", "returns", "cost_returns"):
            assert np.array_equal(_np(tb[k]), g[f"seq_{tag}_{k}"]), (tag, k)  # bit-exact, incl. the fp32 recurrences
        for name, ct in (("prob50", ("affine", -1.0, 50.0)), ("prob8", ("affine", -1.0, 8.0)),
                         ("probinv", ("reciprocal", 10.0))):
            p, cdf = compute_cost_sample_prob(tb, ct, with_cdf=True)
            np.testing.assert_allclose(_np(p), g[f"seq_{tag}_{name}"], rtol=2e-6, atol=1e-9)
            np.testing.assert_allclose(_np(cdf), np.cumsum(g[f"seq_{tag}_{name}"].astype(np.float64)), rtol=0, atol=2e-6)
            assert abs(float(cdf[-1]) - 1.0) < 1e-6


@pytest.mark.parametrize("mode", ["all", "multi-task", "safe", "risky", "boundary"])
def test_ingest_bc_dataset_matches_reference_golden(mode):
    from cases import make_ingest_dataset
    from oracle_util import load_golden
    from osrl_amd.common.ingest import process_bc_dataset
    g = load_golden("ingest")
    for gamma in (1.0, 0.99):
        out = process_bc_dataset(make_ingest_dataset(), 6.0, gamma, mode, DEV)
        tag = f"bc_{mode}_{gamma}"
        assert np.array_equal(_np(out["index"]), g[f"{tag}_index"]), tag
        for k in ("observations", "cost_returns", "rew_returns", "rewards"):
            assert np.array_equal(_np(out[k]), g[f"{tag}_{k}"]), (tag, k)
    with pytest.raises(NotImplementedError):
        process_bc_dataset(make_ingest_dataset(), 6.0, 1.0, "frontier", DEV)


def test_ingest_large_matches_oracle_and_feeds_the_samplers():
    """300k transitions (73 scan tiles, > 1024 episodes, 1-step episodes, no trailing partial episode) against the
    numpy oracle; then the device tables drive SequenceStore / ReplayStore directly."""
    from oracle import ingest_oracle as IO
    from osrl_amd.common.ingest import compute_cost_sample_prob, process_bc_dataset, process_sequence_dataset
    from osrl_amd.common.replay import ReplayStore, SequenceStore
    from osrl_amd.engine.core import StepState
    rs = np.random.RandomState(7)
    n, od, ad = 300_000, 6, 2
    f = np.float32
    data = dict(observations=rs.randn(n, od).astype(f), next_observations=rs.randn(n, od).astype(f),
                actions=rs.uniform(-1, 1, (n, ad)).astype(f), rewards=rs.uniform(0, 1, n).astype(f),
                costs=(rs.uniform(size=n) < 0.1).astype(f), terminals=(rs.uniform(size=n) < 0.004).astype(f),
                timeouts=(rs.uniform(size=n) < 0.004).astype(f))
    data["timeouts"][-1] = 1
    data["terminals"][100:103] = 1  # three 1-step episodes in a row
    tb = process_sequence_dataset(data, False, DEV)
    ref = IO.process_sequence_dataset(data, False)
    assert int(tb["traj_len"].shape[0]) == len(ref) > 1024
    assert np.array_equal(_np(tb["traj_len"]), np.array([len(t["costs"]) for t in ref]))
    for k in ("returns", "cost_returns", "costs"):
        assert np.array_equal(_np(tb[k]), np.concatenate([t[k] for t in ref])), k
    p = compute_cost_sample_prob(tb, ("affine", -1.0, 30.0))
    np.testing.assert_allclose(_np(p), IO.compute_cost_sample_prob(ref, lambda x: 30 - x), rtol=5e-6, atol=1e-10)

    out = process_bc_dataset(data, 20.0, 0.99, "boundary", DEV)
    want = IO.process_bc_dataset(dict(data, index=np.arange(n)), 20.0, 0.99, "boundary")
    assert np.array_equal(_np(out["index"]), want["index"]) and 0 < len(want["index"]) < n
    for k in ("observations", "actions", "cost_returns", "rew_returns", "terminals"):
        assert np.array_equal(_np(out[k]), want[k]), k

    # the device tables feed the samplers without a host round trip
    store = SequenceStore.from_dataset(data, 8, DEV, reward_scale=0.1, cost_scale=1.0, cost_sample=True,
                                       cost_transform=("affine", -1.0, 30.0), seed=3)
    st = StepState(DEV, ["x"])
    st.tick()
    B, T = 256, 8
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
    outs = (z(B, T, od), z(B, T, ad), z(B, T), z(B, T), z(B, T, dt=torch.int64), z(B, T), z(B), z(B, T))
    idx = z(B, 2, dt=torch.int32)
    store.gather(*outs, st.ptr, idx_out=idx)
    torch.cuda.synchronize()
    ii = idx.cpu().numpy()
    o = [x.cpu().numpy() for x in outs]
    for b in range(0, B, 16):
        want_s = prepare_sequence_sample(ref[int(ii[b, 0])], int(ii[b, 1]), T, 0.1, 1.0)
        for got, w in zip(o, want_s):
            np.testing.assert_allclose(got[b], np.asarray(w, np.float64), rtol=1e-6, atol=1e-6)
    safe = process_bc_dataset(data, 20.0, 0.99, "safe", DEV)
    rstore = ReplayStore({k: safe[k] for k in ("observations", "next_observations", "actions", "rewards", "costs",
                                               "terminals", "timeouts")}, DEV)
    assert rstore.n_rows == int(safe["index"].shape[0])


@pytest.mark.parametrize("use_graph,case", [(False, "cdt_small"), (True, "cdt_small"), (True, "cdt_v_prefix_det")])
def test_cdt_batched_evaluate_matches_oracle_rollouts(use_graph, case):
    """CDTTrainer.evaluate on a VecSyntheticSafeEnv (window held in the engine's batch buffers, growing then sliding)
    == the numpy oracle re-slicing a full history per env step (cdt.py:436-518), episode by episode; the episode is
    longer than 2x seq_len and not a multiple of the graph chunk."""
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    from test_gpu_cdt import build_cdt_gpu
    from test_oracle_cdt_golden import build_cdt_oracle
    c = CDT_CASES[case]
    m, tr, lg = build_cdt_gpu(c, use_graph=use_graph)
    o = build_cdt_oracle(c)
    E, EL, T = 5, 2 * c.T + 3, c.T
    m.episode_len = EL
    tr.cost_scale = 2.0
    env = SyntheticSafeEnv(c.od, c.ad, EL, seed=2, init_noise=0.6)
    tr.env = VecSyntheticSafeEnv(env, E, DEV, base_seed=40)
    for rep in range(2):  # the second call replays the cached graph after a reset
        ret, cost, ln = tr.evaluate(E, target_return=30.0, target_cost=5.0)
    rets, costs, lens = tr._rollout[1].run(30.0, 5.0)
    assert abs(ret - rets.mean() / tr.reward_scale) < 1e-5 and ln == EL and abs(cost - costs.mean() / 2.0) < 1e-6
    for e in range(E):
        S, A = np.zeros((EL + 1, c.od), np.float32), np.zeros((EL, c.ad), np.float32)
        R, C = np.zeros(EL + 1, np.float32), np.zeros(EL + 1, np.float32)
        obs, _ = env.reset(seed=40 + e)
        S[0], R[0], C[0] = obs, 30.0, 5.0
        r0 = c0 = 0.0
        for step in range(EL):
            lo = max(0, step + 1 - T)
            n = step + 1 - lo
            pad = lambda x: np.concatenate([x, np.zeros((T - n,) + x.shape[1:], x.dtype)])[None]  # noqa: E731
            acts = o.act_mean(pad(S[lo:step + 1]), pad(A[lo:step + 1]), pad(R[lo:step + 1]), pad(C[lo:step + 1]),
                              pad(np.arange(lo, step + 1)), pad(np.ones(n, np.float32)), np.array([5.0], np.float32))
            act = np.clip(acts[0, n - 1], -1, 1)
            obs, reward, term, trunc, info = env.step(act)
            A[step], S[step + 1] = act, obs
            R[step + 1], C[step + 1] = R[step] - reward, C[step] - info["cost"] * 2.0
            r0 += reward
            c0 += info["cost"]
        assert lens[e] == EL and abs(rets[e] - r0) < 1e-3 * max(1, abs(r0)) and abs(costs[e] - c0) <= 1.0, \
            (e, rets[e], r0, costs[e], c0)
    assert np.unique(np.round(rets, 3)).size == E


def test_replay_store_state_init_and_coptidice_step_replay():
    """ReplayStore(state_init=True) == TransitionDataset(state_init=True) (dataset.py:817-830): is_init = done
    shifted by one with is_init[0] = 1, get_dataset_states(); COptiDICE draws its 7-field minibatch on device."""
    from osrl_amd.algorithms import COptiDICE, COptiDICETrainer
    from osrl_amd.common.logger import DummyLogger
    from osrl_amd.common.replay import ReplayStore, synthetic_transitions
    from osrl_amd.engine.core import StepState
    data = synthetic_transitions(4000, 6, 2, seed=3)
    data["timeouts"][::97] = 1
    store = ReplayStore(data, DEV, reward_scale=0.1, state_init=True, seed=5)
    done = np.logical_or(data["terminals"] == 1, data["timeouts"] == 1).astype(np.float32)
    init = done.copy()
    init[1:] = init[:-1]
    init[0] = 1.0
    p0, ostd, astd = store.get_dataset_states()
    assert abs(p0 - init.mean()) < 1e-7 and ostd.shape == (1, 6) and astd.shape == (1, 2)
    np.testing.assert_allclose(ostd, data["observations"].std(0, keepdims=True), rtol=1e-6)
    B = 512
    st = StepState(DEV, ["x"])
    st.tick()
    z = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    outs = (z(B, 6), z(B, 6), z(B, 2), z(B), z(B), z(B), z(B))
    idx = torch.zeros(B, dtype=torch.int32, device=DEV)
    store.gather(outs, st.ptr, idx_out=idx)
    ii = idx.cpu().numpy()
    np.testing.assert_array_equal(outs[6].cpu().numpy(), init[ii])
    np.testing.assert_array_equal(outs[5].cpu().numpy(), done[ii])
    np.testing.assert_allclose(outs[3].cpu().numpy(), data["rewards"][ii] * 0.1, rtol=1e-6)
    with pytest.raises(ValueError):
        store.gather(outs[:6], st.ptr)

    torch.manual_seed(0)
    m = COptiDICE(6, 2, 1.0, "softchi", p0, ostd, astd, [32, 32], [32, 32], num_nu=2, num_chi=2, device=DEV)
    tr = COptiDICETrainer(m, None, DummyLogger(), 1e-3, 1e-3, 1e-2, device=DEV)
    eng = m.engine(B)
    with pytest.raises(ValueError):
        eng.attach_replay(ReplayStore(data, DEV))
    eng.attach_replay(store)
    p_before = m.groups["nu_network"].p.clone()
    for _ in range(5):
        eng.step_replay()
    torch.cuda.synchronize()
    stats = eng.st.read_stats()
    assert eng.graph is not None and eng.st.device_step() == 5
    assert all(np.isfinite(v) for v in stats.values()) and not torch.equal(p_before, m.groups["nu_network"].p)
    assert abs(m.tau.item() - 1.0) > 1e-3 and abs(m.lmbda.item() - 1.0) > 1e-3


# --------------------------------------------------------------------------- #
# the metric's "cost-return gap vs ref" (SURVEY.md 8c-iii): fixture = the REFERENCE trainers' rollout() loops driving
# the synthetic env with the reference models' act() (tests/golden/make_golden_eval.py)
# --------------------------------------------------------------------------- #
EVAL = dict(episodes=12, episode_len=25, env_seed=1, init_noise=0.7, base_seed=100, cost_scale=2.0)


def _check_gap(name, rets, costs, lens, ref):
    ret_gap = np.abs(rets - ref[:, 0]) / np.maximum(1.0, np.abs(ref[:, 0]))
    assert np.array_equal(lens, ref[:, 2]), name
    assert ret_gap.max() <= 1e-3, (name, ret_gap.max())
    # the cost is an indicator at a threshold: allow one flipped step in at most one episode
    flips = np.abs(costs - ref[:, 1])
    assert (flips > 0).sum() <= 1 and flips.max() <= EVAL["cost_scale"], (name, costs, ref[:, 1])
    print(f"{name}: mean return {rets.mean():.4f} vs ref {ref[:, 0].mean():.4f}; mean cost {costs.mean():.3f} vs "
          f"{ref[:, 1].mean():.3f}")


@pytest.mark.parametrize("name", ["bc_small", "cpq_small", "bcql_small", "bearl_lap", "coptidice_small"])
def test_evaluate_cost_return_gap_vs_reference(name):
    from oracle_util import load_golden
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    from osrl_amd.engine.rollout import BatchedRollout
    g = load_golden("eval_rollouts")
    c = {**CASES, **BEARL_CASES, **COPTIDICE_CASES}[name]
    m, tr, lg = build_gpu(c, use_graph=True)
    E, EL = EVAL["episodes"], EVAL["episode_len"]
    m.episode_len = EL
    cs = EVAL["cost_scale"] if c.algo != "bc" else 1.0
    tr.cost_scale = cs
    env = SyntheticSafeEnv(c.od, c.ad, 50, seed=EVAL["env_seed"], init_noise=EVAL["init_noise"])
    venv = VecSyntheticSafeEnv(env, E, DEV, base_seed=EVAL["base_seed"])
    tr.env = venv
    if c.algo == "bcql":
        ro = BatchedRollout(m, venv, "bcql", cs, z=torch.tensor(g[name + "_z"], device=DEV))
        rets, costs, lens = ro.run()
    else:
        tr.evaluate(E)
        rets, costs, lens = tr._rollout[1].run()
    _check_gap(name, rets, costs, lens, g[name])


@pytest.mark.parametrize("name", ["bc_small", "cpq_small", "bcql_small"])
def test_trained_cost_return_gap_vs_reference(name):
    """The metric's second half AFTER training: ``case.steps`` train steps through the HIP path (seeded batch + injected
    noise = the steps the train-step goldens pin), then the batched on-device evaluate(), against rollouts of the
    REFERENCE models trained by the reference's own train_one_step on the same inputs and rolled out by the reference's
    own rollout() (tests/golden/make_golden_eval_trained.py; cpq.py:294-347)."""
    from gpu_util import gpu_batch, gpu_step
    from oracle_util import load_golden
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    from osrl_amd.engine.rollout import BatchedRollout
    g = load_golden("eval_rollouts_trained")
    c = CASES[name]
    m, tr, lg = build_gpu(c, use_graph=True)
    b = gpu_batch(c)
    assert int(g[name + "_steps"]) == c.steps
    for s in range(c.steps):
        gpu_step(tr, c, b, s)
    E, EL = EVAL["episodes"], EVAL["episode_len"]
    m.episode_len = EL
    cs = EVAL["cost_scale"] if c.algo != "bc" else 1.0
    tr.cost_scale = cs
    env = SyntheticSafeEnv(c.od, c.ad, 50, seed=EVAL["env_seed"], init_noise=EVAL["init_noise"])
    venv = VecSyntheticSafeEnv(env, E, DEV, base_seed=EVAL["base_seed"])
    tr.env = venv
    if c.algo == "bcql":
        rets, costs, lens = BatchedRollout(m, venv, "bcql", cs, z=torch.tensor(g[name + "_z"], device=DEV)).run()
    else:
        tr.evaluate(E)
        rets, costs, lens = tr._rollout[1].run()
    _check_gap(name + " (trained)", rets, costs, lens, g[name])
    # ... and the training moved the policy: the untrained fixture is a different set of episodes
    g0 = load_golden("eval_rollouts")
    assert np.abs(g0[name][:, 0] - g[name][:, 0]).max() > 1e-3


def test_cdt_evaluate_cost_return_gap_vs_reference():
    from oracle_util import load_golden
    from osrl_amd.algorithms import CDT, CDTTrainer
    from osrl_amd.common.logger import DummyLogger
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    g = load_golden("eval_rollouts")
    c = CDT_CASES["cdt_small"]
    E, EL = EVAL["episodes"], EVAL["episode_len"]
    m = CDT(c.od, c.ad, 1.0, seq_len=c.T, episode_len=EL, embedding_dim=c.E, num_layers=c.layers, num_heads=c.heads,
            use_rew=True, use_cost=True, cost_transform=c.cost_transform, stochastic=c.stochastic, init_temperature=0.1,
            target_entropy=-c.ad, device=DEV)
    sd = make_cdt_params(c)
    te, need = sd["timestep_emb.weight"], EL + c.T
    sd["timestep_emb.weight"] = np.concatenate([te] * (need // te.shape[0] + 1))[:need]  # as make_golden_eval.py
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    tr = CDTTrainer(m, None, DummyLogger(), reward_scale=0.1, cost_scale=EVAL["cost_scale"], device=DEV)
    env = SyntheticSafeEnv(c.od, c.ad, 50, seed=EVAL["env_seed"], init_noise=EVAL["init_noise"])
    tr.env = VecSyntheticSafeEnv(env, E, DEV, base_seed=EVAL["base_seed"])
    ret, cost, ln = tr.evaluate(E, target_return=30.0, target_cost=5.0)
    rets, costs, lens = tr._rollout[1].run(30.0, 5.0)
    ref = g["cdt_small"]
    # CDTTrainer.rollout accumulates the RAW cost (cdt.py:513) while the windows carry cost * cost_scale
    saved = EVAL["cost_scale"]
    EVAL["cost_scale"] = 1.0
    try:
        _check_gap("cdt_small", rets, costs, lens, ref)
    finally:
        EVAL["cost_scale"] = saved
    assert abs(ret - ref[:, 0].mean() / 0.1) <= 1e-3 * abs(ref[:, 0].mean() / 0.1)


@pytest.mark.parametrize("pattern", ["none", "all", "single", "last_only", "random_dense"])
def test_ingest_edge_cases(pattern):
    """Episode segmentation / returns / BC selection on degenerate done patterns: no done flag at all (zero complete
    episodes), every transition an episode, a one-transition dataset, one episode covering everything, dense flags."""
    from oracle import ingest_oracle as IO
    from osrl_amd.common.ingest import Episodes, process_bc_dataset, process_sequence_dataset
    rs = np.random.RandomState(11)
    n = 1 if pattern == "single" else 9000  # > 2 scan tiles
    f = np.float32
    term = np.zeros(n, f)
    tout = np.zeros(n, f)
    if pattern == "all":
        tout[:] = 1
    elif pattern in ("single", "last_only"):
        term[-1] = 1
    elif pattern == "random_dense":
        term[rs.uniform(size=n) < 0.3] = 1
        tout[rs.uniform(size=n) < 0.3] = 1
    data = dict(observations=rs.randn(n, 3).astype(f), next_observations=rs.randn(n, 3).astype(f),
                actions=rs.randn(n, 2).astype(f), rewards=rs.uniform(0, 1, n).astype(f),
                costs=(rs.uniform(size=n) < 0.3).astype(f), terminals=term, timeouts=tout)
    ep = Episodes(data, DEV)
    starts, lens = IO.episode_segments(IO.done_flags(data))
    assert ep.n_episodes == len(starts)
    assert np.array_equal(_np(ep.start), starts) and np.array_equal(_np(ep.length), lens)
    tb = process_sequence_dataset(data, True, DEV)
    ref = IO.process_sequence_dataset(data, True)
    if ref:
        for k in ("returns", "cost_returns", "costs", "observations"):
            assert np.array_equal(_np(tb[k]), np.concatenate([t[k] for t in ref])), (pattern, k)
    else:
        assert tb["returns"].numel() == 0 and tb["traj_len"].numel() == 0
    for mode in ("safe", "risky", "multi-task"):
        out = process_bc_dataset(data, 1.0, 0.97, mode, DEV)
        want = IO.process_bc_dataset(dict(data, index=np.arange(n)), 1.0, 0.97, mode)
        assert np.array_equal(_np(out["index"]), want["index"]), (pattern, mode)
        assert np.array_equal(_np(out["observations"]), want["observations"]), (pattern, mode)
        assert np.array_equal(_np(out["cost_returns"]), want["cost_returns"]), (pattern, mode)


def test_ingest_scan_beyond_one_offset_chunk():
    """4.3 M transitions = 1050 scan tiles: the tile-offset pass carries across its 1024-wide chunks; episode
    bounds and the compaction indices stay exact."""
    from oracle import ingest_oracle as IO
    from osrl_amd.common.ingest import Episodes, process_bc_dataset
    rs = np.random.RandomState(3)
    n = 4_300_000
    f = np.float32
    data = dict(terminals=(rs.uniform(size=n) < 0.002).astype(f), timeouts=np.zeros(n, f))
    data["timeouts"][-5] = 1
    ep = Episodes(data, DEV)
    starts, lens = IO.episode_segments(IO.done_flags(data))
    assert ep.n_episodes == len(starts) > 4096
    assert np.array_equal(_np(ep.start), starts) and np.array_equal(_np(ep.length), lens)
    full = dict(data, observations=rs.randn(n, 1).astype(f), next_observations=np.zeros((n, 1), f),
                actions=np.zeros((n, 1), f), rewards=np.ones(n, f), costs=(rs.uniform(size=n) < 0.01).astype(f))
    out = process_bc_dataset(full, 4.0, 1.0, "risky", DEV)
    cr = np.zeros(n, f)
    csum = np.concatenate([[0], np.cumsum(full["costs"], dtype=np.float64)])
    for s, l in zip(starts, lens):
        cr[s:s + l] = csum[s + l] - csum[s]  # gamma = 1: the episode's cost sum (exact in fp32: small integers)
    keep = np.flatnonzero(cr >= 8.0)
    assert np.array_equal(_np(out["index"]), keep) and 0 < len(keep) < n
    assert np.array_equal(_np(out["observations"])[:, 0], full["observations"][keep, 0])


# --------------------------------------------------------------------------- #
# end to end on device: dataset ingestion -> resident store -> training -> batched evaluate -> checkpoint
# --------------------------------------------------------------------------- #
def _expert_dataset(env, n_ep, seed):
    """Episodes of a noisy goal-seeking controller on the synthetic env (DSRL-shaped dict)."""
    rs = np.random.RandomState(seed)
    obs, nobs, act, rew, cost, term, tout = [], [], [], [], [], [], []
    pinv = np.linalg.pinv(env.Bm)
    for e in range(n_ep):
        o, _ = env.reset(seed=1000 + e)
        noise = 0.1 if e % 2 == 0 else 1.5  # careful and sloppy demonstrators: episode costs from 0 to ~10
        for t in range(env.episode_len):
            a = np.clip(pinv @ (env.goal - env.A @ o) + noise * rs.randn(env.action_dim), -1, 1).astype(np.float32)
            o2, r, te, tr_, info = env.step(a)
            obs.append(o); nobs.append(o2); act.append(a); rew.append(r); cost.append(info["cost"])
            term.append(0.0); tout.append(float(tr_))
            o = o2
    f = np.float32
    return dict(observations=np.array(obs, f), next_observations=np.array(nobs, f), actions=np.array(act, f),
                rewards=np.array(rew, f), costs=np.array(cost, f), terminals=np.array(term, f), timeouts=np.array(tout, f))


def test_end_to_end_bc_safe_on_device(tmp_path):
    """BC-Safe as train_bc.py runs it, entirely on device: process_bc_dataset(mode="safe") -> ReplayStore -> BC steps
    drawing their minibatches inside the captured graph -> batched evaluate -> checkpoint round trip.  The cloned
    policy must beat the untrained one on the synthetic env, and the reloaded model must evaluate identically."""
    from osrl_amd.algorithms import BC, BCTrainer
    from osrl_amd.common.checkpoint import load_checkpoint, save_checkpoint
    from osrl_amd.common.ingest import process_bc_dataset
    from osrl_amd.common.logger import DummyLogger
    from osrl_amd.common.replay import ReplayStore
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    od, ad, EL = 6, 2, 30
    env = SyntheticSafeEnv(od, ad, EL, seed=6, init_noise=0.5)
    data = _expert_dataset(env, 120, 0)
    safe = process_bc_dataset(data, cost_limit=1.0, gamma=1.0, bc_mode="safe", device=DEV)
    n_safe = int(safe["index"].shape[0])
    assert 0 < n_safe < len(data["rewards"]) and n_safe % EL == 0  # whole episodes are kept or dropped
    store = ReplayStore({k: safe[k] for k in ("observations", "next_observations", "actions", "rewards", "costs",
                                              "terminals", "timeouts")}, DEV, seed=1)
    torch.manual_seed(0)
    m = BC(od, ad, 1.0, [64, 64], EL, device=DEV)
    tr = BCTrainer(m, None, DummyLogger(), actor_lr=3e-3, device=DEV, stats_mode="none")
    tr.env = VecSyntheticSafeEnv(env, 32, DEV, base_seed=5000)
    before = tr.evaluate(32)
    eng = m.engine(256)
    eng.attach_replay(store)
    for _ in range(400):
        eng.step_replay()
    torch.cuda.synchronize()
    assert eng.graph is not None and eng.st.device_step() == 400
    loss = eng.st.read_stats()["loss/actor_loss"]
    after = tr.evaluate(32)
    assert loss < 0.05 and after[0] > before[0] + 1.0, (loss, before, after)
    path = str(tmp_path / "bc.pt")
    save_checkpoint(m, path)
    torch.manual_seed(1)
    m2 = BC(od, ad, 1.0, [64, 64], EL, device=DEV)
    tr2 = BCTrainer(m2, None, DummyLogger(), actor_lr=3e-3, device=DEV)
    load_checkpoint(m2, path)
    tr2.env = tr.env
    assert tr2.evaluate(32) == after
    with pytest.raises(RuntimeError):
        eng.step(safe["observations"][:256], safe["actions"][:256])


def test_end_to_end_cdt_on_device():
    """CDT as train_cdt.py runs it, entirely on device: SequenceStore.from_dataset (episode split, returns / costs
    to go, cost-weighted trajectory sampling) -> windows drawn inside the captured step -> batched evaluate.  The
    action NLL must fall and the trained policy must reach a clearly better return than the untrained one."""
    from osrl_amd.algorithms import CDT, CDTTrainer
    from osrl_amd.common.logger import DummyLogger
    from osrl_amd.common.replay import SequenceStore
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    od, ad, EL, T = 6, 2, 30, 8
    env = SyntheticSafeEnv(od, ad, EL, seed=6, init_noise=0.5)
    data = _expert_dataset(env, 120, 0)
    store = SequenceStore.from_dataset(data, T, DEV, reward_scale=0.1, cost_scale=1.0, cost_sample=True,
                                       cost_transform=("affine", -1.0, 70.0), seed=2)
    assert store.n_traj == 120 and int(store.traj_len.min()) == EL
    torch.manual_seed(0)
    m = CDT(od, ad, 1.0, seq_len=T, episode_len=EL, embedding_dim=64, num_layers=2, num_heads=4, use_rew=True,
            use_cost=True, cost_transform=True, stochastic=True, init_temperature=0.1, target_entropy=-ad, device=DEV)
    tr = CDTTrainer(m, None, DummyLogger(), learning_rate=2e-3, lr_warmup_steps=20, reward_scale=0.1, cost_scale=1.0,
                    loss_cost_weight=0.02, device=DEV, stats_mode="none")
    tr.env = VecSyntheticSafeEnv(env, 16, DEV, base_seed=7000)
    before = tr.evaluate(16, target_return=0.1 * 25.0, target_cost=2.0)
    eng = m.engine(64, tr.cfg)
    eng.attach_store(store)
    eng.step_store()
    torch.cuda.synchronize()
    first = eng.st.read_stats()["act_loss"]
    for _ in range(300):
        eng.step_store()
    torch.cuda.synchronize()
    last = eng.st.read_stats()
    after = tr.evaluate(16, target_return=0.1 * 25.0, target_cost=2.0)
    assert eng.graph is not None and all(np.isfinite(v) for v in last.values())
    assert last["act_loss"] < first - 0.5, (first, last["act_loss"])
    assert after[0] > before[0] + 10.0 and after[2] == EL, (before, after)


# --------------------------------------------------------------------------------------------------------------- #
# the device minibatch builders against outputs of the REFERENCE's own builders (tests/golden/samples.npz, generated by
# importing SequenceDataset / TransitionDataset: tests/golden/make_golden_ingest.py::make_samples)
# --------------------------------------------------------------------------------------------------------------- #
def test_sequence_store_matches_reference_prepare_sample():
    """SequenceDataset._SequenceDataset__prepare_sample (dataset.py:749-775) on fixed (trajectory, start) pairs: the
    device window gather, fed the same pairs, is bit-exact (slices, tail zero-padding, mask, time_steps, scaling)."""
    from cases import make_ingest_dataset
    from oracle_util import load_golden
    from osrl_amd.common.replay import SequenceStore
    g = load_golden("samples")
    T, RS, CS = 12, 0.1, 2.0
    names = ("states", "actions", "returns", "cost_returns", "time_steps", "mask", "episode_cost", "costs")
    for rev, tag in ((False, "fwd"), (True, "rev")):
        store = SequenceStore.from_dataset(make_ingest_dataset(), T, DEV, reward_scale=RS, cost_scale=CS, cost_reverse=rev)
        pairs = g[f"seq_{tag}_pairs"]
        B, od, ad = len(pairs), store.od, store.ad
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
        outs = (z(B, T, od), z(B, T, ad), z(B, T), z(B, T), z(B, T, dt=torch.int64), z(B, T), z(B), z(B, T))
        idx_in = torch.tensor(pairs, dtype=torch.int32, device=DEV)
        idx_out = z(B, 2, dt=torch.int32)
        store.gather(*outs, None, idx_out=idx_out, idx_in=idx_in)
        torch.cuda.synchronize()
        assert torch.equal(idx_out, idx_in)
        for n, got in zip(names, outs):
            want = g[f"seq_{tag}_{n}"]
            got = got.cpu().numpy()
            assert got.shape == want.shape, (tag, n)
            assert np.array_equal(got.astype(want.dtype), want), (tag, n)


@pytest.mark.parametrize("init", [False, True])
def test_replay_store_matches_reference_prepare_sample(init):
    """TransitionDataset._TransitionDataset__prepare_sample (dataset.py:832-842) on a fixed index list: every golden
    index is among the rows the device sampler draws (65536 draws from 1500 transitions) and its row is bit-exact."""
    from cases import make_ingest_dataset
    from oracle_util import load_golden
    from osrl_amd.common.replay import ReplayStore
    from osrl_amd.engine.core import StepState
    g = load_golden("samples")
    tag = "init" if init else "plain"
    store = ReplayStore(make_ingest_dataset(), DEV, reward_scale=0.1, cost_scale=2.0, seed=5, state_init=init)
    names = ("observations", "next_observations", "actions", "rewards", "costs", "done") + (("is_init",) if init else ())
    B = 65536
    st = StepState(DEV, ["x"])
    st.tick()
    dst = [torch.zeros(B, w, device=DEV) for w in store.widths]
    idx = torch.zeros(B, dtype=torch.int32, device=DEV)
    store.gather(dst, st.ptr, idx_out=idx)
    torch.cuda.synchronize()
    ii = idx.cpu().numpy()
    assert ii.min() >= 0 and ii.max() < store.n_rows
    for j, want_i in enumerate(g[f"trans_{tag}_idx"]):
        rows = np.flatnonzero(ii == want_i)
        assert len(rows), f"index {want_i} never drawn"
        for k, d in zip(names, dst):
            got = d[int(rows[0])].cpu().numpy().reshape(-1)
            want = np.asarray(g[f"trans_{tag}_{k}"][j], np.float32).reshape(-1)
            assert np.array_equal(got, want), (tag, k, int(want_i), got, want)
    if init:
        p, os_, as_ = store.get_dataset_states()
        ref = g["trans_init_states"]
        od = store.widths[0]
        assert abs(p - ref[0]) < 1e-7
        np.testing.assert_allclose(np.ravel(os_), ref[1:1 + od], rtol=1e-6)
        np.testing.assert_allclose(np.ravel(as_), ref[1 + od:], rtol=1e-6)


def test_start_sampling_matches_reference():
    """SequenceDataset(start_sampling=True) (dataset.py:742-744,781-783): the device start-index distribution equals
    compute_start_index_sample_prob of the reference (golden), its per-trajectory cdf ends at 1, and the sampler's
    empirical start histogram follows it."""
    from cases import make_ingest_dataset
    from oracle_util import load_golden
    from osrl_amd.common.ingest import compute_start_index_sample_prob, process_sequence_dataset
    from osrl_amd.common.replay import SequenceStore
    from osrl_amd.engine.core import StepState
    g = load_golden("samples")
    for rev, tag in ((False, "fwd"), (True, "rev")):
        tb = process_sequence_dataset(make_ingest_dataset(), rev, DEV)
        for prob in (0.4, 0.05):
            p, cdf = compute_start_index_sample_prob(tb, prob, with_cdf=True)
            want = g[f"seq_{tag}_startprob_{prob}"]
            np.testing.assert_allclose(_np(p), want, rtol=2e-6, atol=1e-9)
            ends = (_np(tb["traj_start"]) + _np(tb["traj_len"]) - 1).astype(np.int64)
            assert np.abs(_np(cdf)[ends] - 1.0).max() < 1e-6
    T, B = 12, 1 << 17
    store = SequenceStore.from_dataset(make_ingest_dataset(), T, DEV, start_sampling=True, prob=0.4, seed=9)
    st = StepState(DEV, ["x"])
    st.tick()
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
    outs = (z(B, T, store.od), z(B, T, store.ad), z(B, T), z(B, T), z(B, T, dt=torch.int64), z(B, T), z(B), z(B, T))
    idx = z(B, 2, dt=torch.int32)
    store.gather(*outs, st.ptr, idx_out=idx)
    torch.cuda.synchronize()
    ii = idx.cpu().numpy()
    lens, starts = _np(store.traj_len), _np(store.traj_start)
    want = g["seq_fwd_startprob_0.4"]
    tr = int(np.argmax(lens))  # the longest trajectory: its conditional start histogram vs the golden distribution
    sel = ii[ii[:, 0] == tr, 1]
    assert len(sel) > 1500 and sel.min() >= 0 and sel.max() < lens[tr]
    emp = np.bincount(sel, minlength=lens[tr]) / len(sel)
    ref = want[starts[tr]:starts[tr] + lens[tr]]
    assert np.abs(emp - ref).max() < 5 * np.sqrt(ref.max() / len(sel)) + 2e-3
    assert (ii[:, 1] < lens[ii[:, 0]]).all()


# --------------------------------------------------------------------------------------------------------------- #
# the B = 1 .. 8 act() latency path (csrc/act.hip, engine/act.py)
# --------------------------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("name", ["bc_small", "bc_c1", "cpq_small", "cpq_odd", "cpq_wide", "bcql_small", "bcql_pid", "bcql_wide",
                                  "bearl_small", "coptidice_small"])
def test_fast_policy_matches_oracle_and_batched_path(name):
    """model.act(obs) through ONE GEMV kernel on pinned I/O == the oracle policy == the batched fused-MLP path, for
    1 and 4 rows, deterministic and with explicit noise; stays in step with the parameters after train steps."""
    from gpu_util import gpu_batch, gpu_step
    c = {**CASES, **BEARL_CASES, **COPTIDICE_CASES}[name]
    m, tr, lg = build_gpu(c)
    o = build_oracle(c)
    rs = np.random.RandomState(4)
    b = gpu_batch(c)
    for rnd in range(2):
        obs = rs.randn(4, c.od).astype(np.float32)
        if c.algo == "bc":
            want = o.act(obs)
            got1 = np.stack([m.act(obs[i]) for i in range(4)])
            gotn = m.fast_policy().act(obs)[0]
            assert got1.shape == (4, c.ad) and np.abs(got1 - want).max() <= 1e-5 and np.abs(gotn - want).max() <= 1e-5
        elif c.algo == "bcql":
            z = rs.randn(4, 2 * c.ad).astype(np.float32)
            want = np.stack([o.act(obs[i][None], z[i][None])[0] for i in range(4)])
            got1 = np.stack([m.act(obs[i], z=z[i])[0] for i in range(4)])
            assert np.abs(got1 - want).max() <= 1e-5, np.abs(got1 - want).max()
            gotn = m._fast.act(obs, True, noise=z)[0]
            assert np.abs(gotn - want).max() <= 1e-5
            a = np.stack([m.act(obs[0])[0] for _ in range(3)])  # z drawn in the kernel: a fresh draw per call
            assert np.isfinite(a).all() and np.abs(a).max() <= c.max_action + 1e-6 and (a[0] != a[1]).any()
        else:
            want = np.stack([o.act(obs[i][None])[0] for i in range(4)])
            pairs = [m.act(obs[i], True, True) for i in range(4)]
            got1 = np.stack([p[0] for p in pairs])
            assert np.abs(got1 - want).max() <= 1e-5, np.abs(got1 - want).max()
            # log-prob and the stochastic branch against the batched HIP path with the same explicit noise
            from osrl_amd import ops
            eps = rs.randn(4, c.ad).astype(np.float32)
            t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
            ab, lpb = m.actor(t(obs), False, True, eps=t(eps))
            scale = 1.0 if c.algo == "coptidice" else c.max_action
            fp = m._fast
            an, lpn = fp.act(obs, False, noise=eps)
            assert np.abs(an - ab.cpu().numpy() * scale).max() <= 1e-5
            assert np.abs(lpn - lpb.cpu().numpy()).max() <= 1e-4
            lp_det = np.array([float(p[1]) for p in pairs])
            assert np.abs(lp_det - m.actor(t(obs), True, True)[1].cpu().numpy()).max() <= 1e-4
            s = np.stack([m.act(obs[0], False)[0] for _ in range(3)])
            assert np.isfinite(s).all() and (s[0] != s[1]).any(), "stochastic act() must draw fresh noise per call"
        # move the parameters: the latency path reads the canonical weights, nothing to refresh
        for st in range(2):
            gpu_step(tr, c, b, 2 * rnd + st)
            from oracle_util import oracle_step
            oracle_step(o, c, 2 * rnd + st)
