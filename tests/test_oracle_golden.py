"""Pin the oracle (oracle/osrl_oracle.py) against vectors captured from the
reference itself (tests/golden/*.npz, made by tests/golden/make_golden.py).

Gates (SURVEY.md 8c tolerance model): step-1 stats <= 1e-5, <=10-step stats and
parameters <= 1e-4 (observed ~1e-6)."""
import numpy as np
import pytest

from cases import BEARL_CASES, CASES, COPTIDICE_CASES, make_batch

ALL_CASES = {**CASES, **BEARL_CASES, **COPTIDICE_CASES}
from oracle_util import build_oracle, load_golden, oracle_step


@pytest.mark.parametrize("name", list(ALL_CASES))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_matches_reference(name, dtype):
    c = ALL_CASES[name]
    g = load_golden(name)
    keys = [str(k) for k in g["stat_keys"]]
    o = build_oracle(c, dtype)
    for s in range(c.steps):
        st = oracle_step(o, c, s)
        ref = dict(zip(keys, g["stats"][s]))
        tol = 1e-5 if s == 0 else 1e-4
        for k in keys:
            assert abs(st[k] - ref[k]) <= tol * max(1.0, abs(ref[k])), (name, s, k, st[k], ref[k])
        if f"s{s + 1}/log_alpha" in g:
            assert abs(o.log_alpha - float(g[f"s{s + 1}/log_alpha"])) < 1e-6
        if f"s{s + 1}/tau" in g:
            assert abs(o.tau - float(g[f"s{s + 1}/tau"])) < 1e-5 and abs(o.lmbda - float(g[f"s{s + 1}/lmbda"])) < 1e-5
        if f"s{s + 1}/pid_error_old" in g:
            assert abs(o.controller.error_old - float(g[f"s{s + 1}/pid_error_old"])) < 1e-5
            assert abs(o.controller.error_integral - float(g[f"s{s + 1}/pid_error_integral"])) < 1e-5
        for k, v in o.p.items():
            if f"p{s + 1}/{k}" in g:
                np.testing.assert_allclose(v, g[f"p{s + 1}/{k}"], rtol=0, atol=1e-4, err_msg=f"{name} {k}")
            elif f"p{s + 1}/smp/{k}" in g:
                np.testing.assert_allclose(v.reshape(-1)[::97], g[f"p{s + 1}/smp/{k}"], rtol=0, atol=1e-4)
                assert abs(float(v.astype(np.float64).sum()) - float(g[f"p{s + 1}/sum/{k}"])) < 1e-3
    # Adam moments of the first tensor of each optimizer
    for oname, opt in (("actor_optim", getattr(o, "opt_actor", getattr(o, "opt", None))),
                       ("critic_optim", getattr(o, "opt_critic", None)),
                       ("cost_critic_optim", getattr(o, "opt_cost", None)),
                       ("nu_optim", getattr(o, "opt_nu", None)), ("chi_optim", getattr(o, "opt_chi", None)),
                       ("vae_optim", getattr(o, "opt_vae", None))):
        if opt is None or f"adam/{oname}/exp_avg" not in g:
            continue
        m_ref = g[f"adam/{oname}/exp_avg"]
        k0 = [k for k in opt.keys if opt.m[k].shape == m_ref.shape][0]
        assert opt.t == int(g[f"adam/{oname}/step"])
        np.testing.assert_allclose(opt.m[k0], m_ref, rtol=0, atol=1e-6)
        np.testing.assert_allclose(opt.v[k0], g[f"adam/{oname}/exp_avg_sq"], rtol=0, atol=1e-6)
    # act()
    b = make_batch(c)
    if c.algo == "bcql":
        a = o.act(b["observations"], g["act_z"])
    else:
        a = o.act(b["observations"])
    np.testing.assert_allclose(a, g["act"], rtol=0, atol=1e-4)


# --------------------------------------------------------------------------- #
# dataset ingestion oracle (oracle/ingest_oracle.py) vs the reference's own functions (tests/golden/ingest.npz)
# --------------------------------------------------------------------------- #
def test_ingest_oracle_matches_golden():
    from cases import make_ingest_dataset
    from oracle import ingest_oracle as IO
    g = load_golden("ingest")
    for rev, tag in ((False, "fwd"), (True, "rev")):
        traj = IO.process_sequence_dataset(make_ingest_dataset(), rev)
        assert np.array_equal(np.array([len(t["costs"]) for t in traj]), g[f"seq_{tag}_len"])
        for k in ("observations", "actions", "rewards", "costs", "returns", "cost_returns"):
            got = np.concatenate([t[k] for t in traj])
            assert got.dtype == g[f"seq_{tag}_{k}"].dtype == np.float32
            assert np.array_equal(got, g[f"seq_{tag}_{k}"]), (tag, k)  # bit-exact: same fp32 recurrence
        for name, fn in (("prob50", lambda x: 50 - x), ("prob8", lambda x: 8 - x), ("probinv", lambda x: 1 / (x + 10))):
            np.testing.assert_allclose(IO.compute_cost_sample_prob(traj, fn), g[f"seq_{tag}_{name}"], rtol=1e-6, atol=0)
    for mode in IO.BC_MODES:
        for gamma in (1.0, 0.99):
            data = make_ingest_dataset()
            data["index"] = np.arange(data["rewards"].shape[0])
            out = IO.process_bc_dataset(data, 6.0, gamma, mode)
            tag = f"bc_{mode}_{gamma}"
            assert np.array_equal(out["index"], g[f"{tag}_index"]), tag
            for k in ("observations", "cost_returns", "rew_returns", "rewards"):
                assert np.array_equal(out[k], g[f"{tag}_{k}"]), (tag, k)
    with pytest.raises(NotImplementedError):
        IO.process_bc_dataset(make_ingest_dataset(), 6.0, 1.0, "frontier")


def test_minibatch_builders_match_reference_samples():
    """oracle.prepare_sequence_sample / transition_sample / compute_start_index_sample_prob vs outputs of the
    reference's OWN SequenceDataset.__prepare_sample, TransitionDataset.__prepare_sample and
    compute_start_index_sample_prob (tests/golden/samples.npz, generated by importing the reference)."""
    from cases import make_ingest_dataset
    from oracle import ingest_oracle as IO
    from oracle.osrl_oracle import prepare_sequence_sample, transition_sample
    g = load_golden("samples")
    T, RS, CS = 12, 0.1, 2.0
    names = ("states", "actions", "returns", "cost_returns", "time_steps", "mask", "episode_cost", "costs")
    for rev, tag in ((False, "fwd"), (True, "rev")):
        traj = IO.process_sequence_dataset(make_ingest_dataset(), rev)
        for j, (tr, st) in enumerate(g[f"seq_{tag}_pairs"]):
            got = prepare_sequence_sample(traj[int(tr)], int(st), T, RS, CS)
            for n, v in zip(names, got):
                want = g[f"seq_{tag}_{n}"][j]
                assert np.asarray(v).shape == want.shape, (tag, n, j)
                assert np.array_equal(np.asarray(v, want.dtype), want), (tag, n, int(tr), int(st))
        for prob in (0.4, 0.05):
            got = np.concatenate(IO.compute_start_index_sample_prob(traj, prob))
            np.testing.assert_allclose(got, g[f"seq_{tag}_startprob_{prob}"], rtol=1e-12, atol=0)
    data = make_ingest_dataset()
    for tag in ("plain", "init"):
        idx = g[f"trans_{tag}_idx"]
        got = transition_sample(data, idx, RS, CS)
        for k, v in zip(("observations", "next_observations", "actions", "rewards", "costs", "done"), got):
            assert np.array_equal(np.asarray(v, np.float32), g[f"trans_{tag}_{k}"].astype(np.float32)), (tag, k)


@pytest.mark.parametrize("kernel", ["gaussian", "laplacian"])
def test_bearl_oracle_mmd_gradient_by_finite_differences(kernel):
    """The hand-derived d MMD / d y of oracle/bearl_oracle.py against central differences (fp64), independent of the
    goldens."""
    from oracle.bearl_oracle import mmd_and_grad
    rs = np.random.RandomState(0)
    B, M, d = 3, 5, 4
    x, y = rs.randn(B, M, d), rs.randn(B, M, d)
    _, g = mmd_and_grad(x, y, 1.7, kernel)
    h = 1e-6
    for (b, j, k) in [(0, 0, 0), (1, 3, 2), (2, 4, 3), (0, 2, 1)]:
        yp, ym = y.copy(), y.copy()
        yp[b, j, k] += h
        ym[b, j, k] -= h
        fd = (mmd_and_grad(x, yp, 1.7, kernel)[0][b] - mmd_and_grad(x, ym, 1.7, kernel)[0][b]) / (2 * h)
        assert abs(fd - g[b, j, k]) < 1e-6 * max(1.0, abs(fd)), (kernel, b, j, k, fd, g[b, j, k])


def test_torch_cpu_baseline_matches_numpy_oracle():
    """oracle/torch_cpq_cpu.py (bench.py's torch-on-CPU baseline, autograd + torch.optim.Adam) against the numpy
    oracle (hand-derived backward, pinned to the reference's golden vectors above): same parameters after 3 steps,
    same logged losses -- so the thing bench.py times as "the reference's CPU path" computes the reference's step."""
    import torch
    from cases import CASES, hyper, make_batch, make_noise, make_params
    from oracle.torch_cpq_cpu import TorchCPQ
    from oracle_util import build_oracle, oracle_step
    c = CASES["cpq_small"]
    hp = hyper(c)
    o = build_oracle(c)
    t = TorchCPQ(make_params(c), max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"], tau=hp["tau"],
                 beta=hp["beta"], qc_scalar=hp["qc_scalar"], cost_limit=c.cost_limit, episode_len=c.episode_len,
                 actor_lr=hp["actor_lr"], critic_lr=hp["critic_lr"], alpha_lr=hp["alpha_lr"], vae_lr=hp["vae_lr"])
    b = make_batch(c)
    for step in range(3):
        so = oracle_step(o, c, step)
        st = t.train_one_step(b["observations"], b["next_observations"], b["actions"], b["rewards"], b["costs"],
                              b["done"], make_noise(c, step))
        for k in so:
            assert abs(so[k] - st[k]) <= 2e-5 * max(1.0, abs(so[k])), (step, k, so[k], st[k])
    for k, v in o.p.items():
        d = float(np.abs(v - t.p[k].detach().numpy()).max())
        assert d <= 2e-5 * max(1.0, float(np.abs(v).max())), (k, d)


@pytest.mark.parametrize("name", ["bc_small", "bcql_small", "bcql_pid", "cdt_small", "cdt_det", "cdt_mid"])
def test_torch_cpu_baselines_match_reference_goldens(name):
    """oracle/torch_cpu_baselines.py (bench.py's torch-on-CPU baselines for C1 / C3 / C5: autograd + torch.optim) pinned
    DIRECTLY to the vectors captured from the reference: every step's logged statistics and the parameters at the
    snapshot steps -- so what bench.py times as "the reference's CPU path" of those configs computes the reference's step
    (bc.py:103-109, bcql.py:283-306, cdt.py:343-418)."""
    import torch
    from cases import CASES, CDT_CASES, hyper, make_batch, make_cdt_batch, make_cdt_params, make_noise, make_params
    from oracle.torch_cpu_baselines import TorchBC, TorchBCQL, TorchCDT
    from oracle_util import load_golden
    torch.manual_seed(0)
    g = load_golden(name)
    keys = [str(k) for k in g["stat_keys"]]
    if name.startswith("cdt"):
        c = CDT_CASES[name]
        t = TorchCDT(make_cdt_params(c), seq_len=c.T, num_heads=c.heads, num_layers=c.layers,
                     cost_transform=c.cost_transform, stochastic=c.stochastic, init_temperature=0.1, target_entropy=-c.ad,
                     learning_rate=c.lr, weight_decay=c.wd, clip_grad=c.clip, lr_warmup_steps=c.warmup,
                     loss_cost_weight=c.cost_w, loss_state_weight=c.state_w)
        b = make_cdt_batch(c)
        step = lambda s: t.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"],  # noqa: E731
                                          b["mask"], b["episode_cost"], b["costs"])
    else:
        c = CASES[name]
        hp = hyper(c)
        b = make_batch(c)
        if c.algo == "bc":
            t = TorchBC(make_params(c), c.max_action, hp["actor_lr"])
            step = lambda s: t.train_one_step(b["observations"], b["actions"])  # noqa: E731
        else:
            t = TorchBCQL(make_params(c), max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"], tau=hp["tau"],
                          phi=hp["phi"], lmbda=hp["lmbda"], beta=hp["beta"], PID_gains=hp["PID"], cost_limit=c.cost_limit,
                          episode_len=c.episode_len, actor_lr=hp["actor_lr"], critic_lr=hp["critic_lr"], vae_lr=hp["vae_lr"])
            step = lambda s: t.train_one_step(b["observations"], b["next_observations"], b["actions"], b["rewards"],  # noqa: E731
                                              b["costs"], b["done"], make_noise(c, s))
    for s in range(c.steps):
        st = step(s)
        ref = dict(zip(keys, g["stats"][s]))
        tol = 1e-5 if s == 0 else 1e-4
        for k in keys:
            assert abs(st[k] - ref[k]) <= tol * max(1.0, abs(ref[k])), (name, s, k, st[k], ref[k])
        if f"s{s + 1}/log_temperature" in g:
            assert abs(t.log_temperature.item() - float(g[f"s{s + 1}/log_temperature"])) < 1e-6
        if f"s{s + 1}/pid_error_old" in g:
            assert abs(t.error_old - float(g[f"s{s + 1}/pid_error_old"])) < 1e-5
            assert abs(t.error_integral - float(g[f"s{s + 1}/pid_error_integral"])) < 1e-5
        for k, v in t.p.items():
            a = v.detach().numpy()
            if f"p{s + 1}/{k}" in g:
                np.testing.assert_allclose(a, g[f"p{s + 1}/{k}"], rtol=0, atol=2e-5, err_msg=f"{name} step {s + 1} {k}")
            elif f"p{s + 1}/smp/{k}" in g:
                np.testing.assert_allclose(a.reshape(-1)[::97], g[f"p{s + 1}/smp/{k}"], rtol=0, atol=2e-5)
