"""Data-parallel correctness at world_size 2 on ONE GPU: two replicas run the sharded step in two host threads (own
streams, eager launches) and exchange through an in-process stand-in for the process group that implements the two
primitives the engines use (all-reduce SUM, all-gather) with exactly torch.distributed's semantics.  Oracle
(SURVEY.md 8e): the sharded step on 2 x B rows == the single-device step on the concatenated 2B-row batch.

This covers what the 1-rank NCCL tests cannot (there every collective is an identity): the 1/B_global normalisations,
the batch-global statistics (CPQ's quantile and OOD mean, the PID / dual-variable means of BCQ-L and BEAR-L) and the
logged statistics' shares."""
import threading

import numpy as np
import pytest
import torch

from cases import Case, make_batch, make_noise
from gpu_util import build_gpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Shared:
    def __init__(self, world):
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)
        self.errors = []


def make_sim(shared, rank):
    from osrl_amd.engine.dist import DataParallel

    class SimDist(DataParallel):
        def __init__(self):  # no process group: the exchange happens through `shared`
            self.group, self.world, self.rank, self._gather_buf = None, shared.world, rank, None

        def _exchange(self, t):
            torch.cuda.current_stream().synchronize()
            shared.slots[rank] = t
            shared.barrier.wait()
            parts = [s.clone() for s in shared.slots]
            torch.cuda.current_stream().synchronize()
            shared.barrier.wait()  # everyone holds private copies: the originals may be overwritten now
            return parts

        def all_reduce_(self, t):
            parts = self._exchange(t)
            acc = parts[0]
            for p in parts[1:]:
                acc = acc + p
            t.copy_(acc)
            return t

        def all_gather_concat(self, t):
            return torch.cat([p.reshape(-1) for p in self._exchange(t)])

    return SimDist()


def _shard(v, key, r, W, B, N, M=1):
    """Rows of rank r out of a concatenated-batch tensor; noise tensors keep their per-algorithm row order."""
    Bl = B // W
    if key == "eps_ood":  # [N, B, ad]
        return v[:, r * Bl:(r + 1) * Bl].copy()
    if key == "eps_vae_ood":  # [N*B, L], row j*B + b (sample-major, cpq.py:170-176)
        return v.reshape(N, B, -1)[:, r * Bl:(r + 1) * Bl].reshape(N * Bl, -1).copy()
    if key in ("z_c", "z_cc", "eps_c", "eps_cc"):  # [N*B, .], row b*N + j (repeat_interleave)
        return v.reshape(B, N, -1)[r * Bl:(r + 1) * Bl].reshape(Bl * N, -1).copy()
    if key == "z_mmd":  # [B, M, L]
        return v[r * Bl:(r + 1) * Bl].copy()
    if key == "eps_pi":  # [B*M, ad], row b*M + j
        return v.reshape(B, M, -1)[r * Bl:(r + 1) * Bl].reshape(Bl * M, -1).copy()
    return v[r * Bl:(r + 1) * Bl].copy()  # per-row tensors


DP_CASES = {
    "cpq": Case("dp_cpq", "cpq", od=5, ad=2, B=32, hidden=[32, 32], vae_hidden=48, N=4, steps=3, episode_len=1000, seed=7),
    "bcql": Case("dp_bcql", "bcql", od=4, ad=2, B=32, hidden=[24, 24], vae_hidden=32, N=3, num_q=1, num_qc=2, steps=3,
                 episode_len=200, cost_limit=-4.0, max_action=1.5, seed=8),  # PID active
    "bearl": Case("dp_bearl", "bearl", od=4, ad=2, B=32, hidden=[24, 24], vae_hidden=32, N=3, num_q=1, num_qc=2, steps=3,
                  episode_len=200, cost_limit=-4.0, seed=9, hp=dict(M=4, kernel="laplacian", mmd_sigma=1.5, alpha_lr=0.05)),
    "bc": Case("dp_bc", "bc", od=8, ad=2, B=32, hidden=[32, 32], steps=3, seed=10),
    # COptiDICE: the chi loss takes a softmax over the GLOBAL batch (all-gathered ell); with and without the chi net
    "coptidice": Case("dp_dice", "coptidice", od=5, ad=2, B=32, hidden=[24, 24], steps=3, episode_len=200, seed=11,
                      hp=dict(actor_lr=1e-3, critic_lr=1e-3, scalar_lr=1e-2)),
    "coptidice_nochi": Case("dp_dice0", "coptidice", od=5, ad=2, B=32, hidden=[24, 24], num_q=1, num_qc=2, steps=3,
                            episode_len=200, cost_limit=40.0, seed=12,
                            hp=dict(f_type="kl", cost_ub_epsilon=0.0, actor_lr=1e-3, critic_lr=1e-3, scalar_lr=1e-2)),
}


@pytest.mark.parametrize("algo", list(DP_CASES))
def test_world2_sharded_step_equals_concatenated_batch(algo):
    c = DP_CASES[algo]
    W, B, N, M = 2, c.B, c.N, int(c.hp.get("M", 1))
    Bl = B // W
    batch = make_batch(c)
    keys = ("observations", "actions") if algo == "bc" else \
        ("observations", "next_observations", "actions", "rewards", "costs", "done") + \
        (("is_init",) if c.algo == "coptidice" else ())
    step = (lambda tr, args, nz: tr.train_one_step(list(args), noise=nz)) if c.algo == "coptidice" else \
        (lambda tr, args, nz: tr.train_one_step(*args) if algo == "bc" else tr.train_one_step(*args, noise=nz))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731

    # single device, concatenated batch
    m1, tr1, lg1 = build_gpu(c)
    for s in range(c.steps):
        nz = {k: t(v) for k, v in make_noise(c, s).items()}
        step(tr1, [t(batch[k]) for k in keys], nz)
    torch.cuda.synchronize()

    # two replicas, each on its half
    shared = _Shared(W)
    reps = [build_gpu(c) for _ in range(W)]

    def worker(r):
        try:
            m, tr, lg = reps[r]
            with torch.cuda.stream(torch.cuda.Stream()):
                sim = make_sim(shared, r)
                m.engine(Bl, rows_global=B, dist=sim)
                args = [t(batch[k][r * Bl:(r + 1) * Bl]) for k in keys]
                for s in range(c.steps):
                    nz = {k: t(_shard(v, k, r, W, B, N, M)) for k, v in make_noise(c, s).items()}
                    step(tr, args, nz)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:  # noqa: BLE001
            shared.errors.append((r, repr(e)))
            shared.barrier.abort()
            raise

    th = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=120)
    assert not shared.errors, shared.errors
    torch.cuda.synchronize()

    sd1 = {k: v.detach().cpu().numpy() for k, v in m1.state_dict().items()}
    for r in range(W):
        m, tr, lg = reps[r]
        for k, v in m.state_dict().items():
            d = np.abs(v.detach().cpu().numpy() - sd1[k]).max()
            assert d <= 2e-5, f"{algo} rank {r} param {k}: sharded vs concatenated differ by {d:.3e}"
        for name in ("log_alpha", "pid_state", "scalar_leaves"):
            if hasattr(m, name):
                a, b = getattr(m, name).cpu().numpy(), getattr(m1, name).cpu().numpy()
                assert np.abs(a - b).max() <= 1e-5, (algo, r, name, a, b)
        # logged statistics: the all-reduced values equal the single-device ones on every rank
        for k, vals in lg1.data.items():
            got = [float(x) for x in lg.data[k]]
            want = [float(x) for x in vals]
            assert np.allclose(got, want, rtol=1e-4, atol=1e-5), (algo, r, k, got, want)
    # the replicas stay identical to each other (bit for bit: same reduced gradients, same optimizer)
    for k, v in reps[0][0].state_dict().items():
        assert torch.equal(v, reps[1][0].state_dict()[k]), f"{algo}: replicas diverged in {k}"


def test_world2_cdt_sharded_step_equals_concatenated_batch():
    """CDT: the count-normalised means ([mask > 0].mean(), accuracy) use GLOBAL counts (the two shards hold different
    numbers of valid tokens), the gradient is clipped by the norm of the REDUCED gradient, the temperature step
    sees the global entropy."""
    from cases import CDT_CASES, make_cdt_batch
    from test_gpu_cdt import build_cdt_gpu
    c = CDT_CASES["cdt_small"]
    W, B = 2, c.B
    Bl = B // W
    batch = make_cdt_batch(c)
    batch["mask"][0, 1:] = 0  # more (tail) padding in rank 0's shard: the valid-token counts of the shards differ
    assert len({float(batch["mask"][r * Bl:(r + 1) * Bl].sum()) for r in range(W)}) == W, "shards must differ in valid tokens"
    keys = ("states", "actions", "returns", "costs_return", "time_steps", "mask", "episode_cost", "costs")
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731
    m1, tr1, lg1 = build_cdt_gpu(c)
    for s in range(3):
        tr1.train_one_step(*[t(batch[k]) for k in keys])
    torch.cuda.synchronize()

    shared = _Shared(W)
    reps = [build_cdt_gpu(c) for _ in range(W)]

    def worker(r):
        try:
            m, tr, lg = reps[r]
            with torch.cuda.stream(torch.cuda.Stream()):
                m.engine(Bl, tr.cfg, dist=make_sim(shared, r))
                args = [t(batch[k][r * Bl:(r + 1) * Bl]) for k in keys]
                for s in range(3):
                    tr.train_one_step(*args)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:  # noqa: BLE001
            shared.errors.append((r, repr(e)))
            shared.barrier.abort()
            raise

    th = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=120)
    assert not shared.errors, shared.errors
    torch.cuda.synchronize()
    sd1 = m1.state_dict()
    for r in range(W):
        m, tr, lg = reps[r]
        for k, v in m.state_dict().items():
            if v.dtype != torch.bool:
                d = float((v - sd1[k]).abs().max())
                assert d <= 2e-6, f"cdt rank {r} param {k}: {d:.3e}"
        assert abs(float(m.log_temperature) - float(m1.log_temperature)) < 1e-6
        for k, vals in lg1.data.items():
            assert np.allclose([float(x) for x in lg.data[k]], [float(x) for x in vals], rtol=1e-4, atol=1e-5), (r, k)
