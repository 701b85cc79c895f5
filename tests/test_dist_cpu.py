"""world_size-2 gloo test (CPU) of the data-parallel reduction algebra used by osrl_amd/engine/dist.py:
a step sharded over W ranks with 1/B_global normalisation + all-reduce(SUM) of gradients, all-gather
for the batch-global quantile and a scalar all-reduce equals the single-device computation on the
concatenated batch.  Compute is the numpy oracle (CPU); the collectives are the real
DataParallel methods."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cases import CASES, make_batch, make_noise, make_params
        from oracle.osrl_oracle import VAE, quantile_linear
        from osrl_amd.engine.dist import DataParallel
        dp = DataParallel()
        assert dp.world == world and dp.rank == rank
        c = CASES["cpq_small"]
        p = {k: v.astype(np.float64) for k, v in make_params(c).items()}
        b, nz = make_batch(c), make_noise(c, 0)
        Bg = c.B
        sl = slice(rank * Bg // world, (rank + 1) * Bg // world)
        vae = VAE(c.max_action)
        obs, act, eps = (b["observations"].astype(np.float64), b["actions"].astype(np.float64),
                         nz["eps_vae"].astype(np.float64))
        # full-batch reference
        loss_full, g_full = vae.loss_and_grads(p, obs, act, eps, 0.5)
        # shard: local mean-loss gradient * (B_local / B_global) == gradient with 1/B_global normalisation
        loss_loc, g_loc = vae.loss_and_grads(p, obs[sl], act[sl], eps[sl], 0.5)
        keys = sorted(g_full)
        flat = torch.from_numpy(np.concatenate([g_loc[k].reshape(-1) for k in keys]) / world)
        dp.all_reduce_(flat)
        ref = np.concatenate([g_full[k].reshape(-1) for k in keys])
        assert np.abs(flat.numpy() - ref).max() < 1e-12
        stat = torch.tensor([loss_loc / world])
        dp.all_reduce_(stat)
        assert abs(stat.item() - loss_full) < 1e-12
        # batch-global quantile through all-gather
        rs = np.random.RandomState(5)
        kl = rs.randn(world, 4 * 6).astype(np.float32)
        allv = dp.all_gather_concat(torch.from_numpy(kl[rank].copy()))
        assert allv.numel() == kl.size
        assert quantile_linear(allv.numpy(), 0.75) == quantile_linear(kl.reshape(-1), 0.75)
        # coalesced all-reduce of several tensors (CPQ: [critic grads | cost grads | qc_ood mean], [actor | stats]);
        # on gloo it degrades to one all-reduce per tensor, same result
        ts = [torch.full((5,), float(rank + 1)), torch.tensor([10.0 * (rank + 1)]), None, torch.ones(3) * rank]
        dp.all_reduce_many_(ts)
        assert torch.equal(ts[0], torch.full((5,), 3.0)) and float(ts[1]) == 30.0 and torch.equal(ts[3], torch.ones(3))
        # broadcast_model-style broadcast
        t = torch.full((4,), float(rank))
        dist.broadcast(t, src=0)
        assert float(t.sum()) == 0.0
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_algebra_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
