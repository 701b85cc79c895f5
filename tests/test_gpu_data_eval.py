"""GPU tests of the minibatch sources (on-device CDT window sampler vs the numpy restatement of
SequenceDataset.__prepare_sample) and of evaluate()/rollout()/act() against the oracle policy on the
build-owned synthetic environment."""
import numpy as np
import pytest
import torch

from cases import CASES, CDT_CASES, make_cdt_params
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


def test_cdt_rollout_matches_oracle():
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv
    from test_gpu_cdt import build_cdt_gpu
    from test_oracle_cdt_golden import build_cdt_oracle
    c = CDT_CASES["cdt_small"]
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
                          pad(np.arange(lo, step + 1)), mask)
        act = np.clip(acts[0, n - 1], -1, 1)
        obs, reward, term, trunc, info = env_o.step(act)
        A[step], S[step + 1] = act, obs
        R[step + 1], C[step + 1] = R[step] - reward, C[step] - info["cost"]
        r0 += reward
        c0 += info["cost"]
    assert ln == EL and abs(ret - r0 / tr.reward_scale) < 1e-3 * max(1, abs(r0)) and abs(cost - c0) < 0.5
