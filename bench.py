#!/usr/bin/env python3
"""bench.py -- grad-steps/sec of the OSRL train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5]

N > 1 without a torchrun environment re-executes itself under ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1`` (one rank per GPU, RCCL); launched BY torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment as usual.

Default workload (BASELINE.json configs[1], the one the metric is quoted on): c2 = CPQ, (obs_dim, act_dim) =
(76, 2) [OfflinePointGoal1], batch 2048 per GPU, hidden [256,256], VAE 400, N=10 sampled actions, num_q = num_qc = 2,
fp32, train-config learning rates -- every phase of CPQTrainer.train_one_step (vae, critic, cost-critic incl. the
N*B OOD scoring + quantile, actor, Adam x4, Polyak x3) plus the on-device minibatch draw from a HBM-resident
synthetic transition store and the Gaussian noise generation are INSIDE the timed step.  The other configs of
BASELINE.json run with --config: c1 BC (8,2) B=256, c3 BCQ-Lag (33,8) B=4096, c4 CPQ (17,6) 2048 rows per GPU
(global 16384 at 8 GPUs), c5 CDT (11,3) T=20 E=256 8 heads 3 layers B=1024.

Unit: one step = one gradient step on one GPU's batch.  With N GPUs the job is data parallel (global batch = N x
per-GPU batch, one optimizer step per iteration, gradient all-reduce over RCCL), so ``value`` = N x optimizer-steps/s
= per-GPU-batch gradient steps per second over the whole job ("scaling": "weak"); ``optimizer_steps_per_s`` and
``transitions_per_s`` are printed beside it.

Timing: probes -> [W warm-up steps -> barrier + synchronize -> EXACTLY K steps -> barrier + synchronize] reported as
``no_preroll`` -> ``--preroll-ms`` (40) of untimed GEMM work (the device's power state needs ~25 ms of load after any idle
gap, see preroll()) -> the same W + K sequence again = ``value``; max over ranks.  Both protocols are in every line.

Prints ONE JSON line (rank 0) with the driver's contract plus ``roofline`` and ``cpu_baseline``.
Nothing here reads /root/reference.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0
HID, VAE_H, NS = [256, 256], 400, 10

# BASELINE.json configs -> shapes ("Config shorthand")
CONFIGS = {
    "c1": dict(algo="bc", od=8, ad=2, B=256, episode_len=300,
               desc="BC-Safe train_one_step, OfflineCarCircle-shaped (obs 8, act 2), batch 256/GPU, MLP [256,256]"),
    "c2": dict(algo="cpq", od=76, ad=2, B=2048, episode_len=1000,
               desc="CPQ train_one_step, OfflinePointGoal1-shaped (obs 76, act 2), batch 2048/GPU, hidden [256,256], "
                    "VAE 400, N=10, num_q=num_qc=2"),
    "c3": dict(algo="bcql", od=33, ad=8, B=4096, episode_len=200,
               desc="BCQ-Lag train_one_step, OfflineAntRun-shaped (obs 33, act 8), batch 4096/GPU, hidden [256,256], "
                    "VAE 400, N=10, twin critics + twin cost critics"),
    "c4": dict(algo="cpq", od=17, ad=6, B=2048, episode_len=1000,
               desc="CPQ train_one_step, OfflineHalfCheetah-shaped (obs 17, act 6), batch 2048/GPU (global 16384 at 8 "
                    "GPUs), hidden [256,256], VAE 400, N=10, num_q=num_qc=2"),
    "c5": dict(algo="cdt", od=11, ad=3, B=1024, episode_len=1000, T=20, E=256, heads=8, layers=3,
               desc="CDT train_one_step, OfflineHopper-shaped (obs 11, act 3), seq_len 20, embed 256, 8 heads, 3 layers, "
                    "dropout 0.1, batch 1024/GPU"),
}


def lin(sizes):
    return sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))


def flops_per_step(cfg) -> float:
    """Algorithmic FLOPs of one reference step on one GPU's batch (SURVEY.md 8d formulas; 1 MAC = 2 FLOP)."""
    od, ad, Bsz, H, V, N = cfg["od"], cfg["ad"], cfg["B"], HID, VAE_H, NS
    if cfg["algo"] == "bc":
        return 2.0 * 3 * lin([od] + H + [ad]) * Bsz
    if cfg["algo"] == "cpq":
        nq = nqc = 2
        actor = lin([od] + H) + 2 * H[-1] * ad
        q = lin([od + ad] + H + [1])
        vae = lin([od + ad, V, V]) + 2 * V * 2 * ad + lin([od + 2 * ad, V, V, ad])
        step = 3 * vae + (3 * nq * q + actor + nq * q + nqc * q) + \
            (3 * nqc * q + 2 * actor + nqc * q + N * nqc * q + N * vae) + (3 * actor + 2 * nq * q + nqc * q)
        return 2.0 * step * Bsz
    if cfg["algo"] == "bcql":
        nq = nqc = 2
        actor = lin([od + ad] + H + [ad])
        q = lin([od + ad] + H + [1])
        dec = lin([od + 2 * ad, V, V, ad])
        vae = lin([od + ad, V, V]) + 2 * V * 2 * ad + dec
        step = 3 * vae + 2 * (3 * 2 * nq * q + N * (dec + actor + 2 * nq * q)) + (dec + 3 * actor + 2 * 2 * nq * q + 2 * 2 * nqc * q)
        return 2.0 * step * Bsz
    T, E, Lyr = cfg["T"], cfg["E"], cfg["layers"]
    per = 3 * (Lyr * (12 * E * E + 2 * (4 * T) * E) * 4 * T + T * (od + ad + 2) * E + T * E * (2 * ad + od + 2))
    return 2.0 * per * Bsz


def executed_flops_per_step(cfg, ood_rows: bool = False, share=(0, 0)) -> float:
    """FLOPs this build actually ISSUES per step.  The reference's step (flops_per_step) contains work whose result
    it discards or computes twice; the engine skips exactly that (same results):
      * CPQ ``cost_critic_loss`` runs the whole VAE on the N*B sampled actions and keeps only the KL of the ENCODER
        output (cpq.py:176-182: ``_, _, mean, std = self.vae(...)``): the decoder on N*B rows is never launched;
      * the actor trunk on next_obs (cpq.py:141 and :159) and on obs (:164 and :209) is evaluated once each;
      * ``ood_rows`` (the engine's plan.ood_rows on one GPU): the target cost critics on the quarter of the N*B sampled
        actions that enters ``qc_ood`` (cpq.py:183-184) instead of all of them;
      * ``share`` = (k16 of the target cost critics' launch, k16 of the encoder's; plan.ood_share): on tiles of shared
        observations the first 16 k16 input columns of layer 0 are multiplied once per observation of a tile (5 copies each)
        instead of once per row (osrl_rows_t.share0).
    Other algorithms: nothing skipped."""
    if cfg["algo"] != "cpq":
        return flops_per_step(cfg)
    od, ad, Bsz, H, V, N = cfg["od"], cfg["ad"], cfg["B"], HID, VAE_H, NS
    actor = lin([od] + H) + 2 * H[-1] * ad
    dec = lin([od + 2 * ad, V, V, ad])
    fx = flops_per_step(cfg) - 2.0 * (N * dec + 2 * actor) * Bsz
    if ood_rows:
        # plan.ood_rows (round 6): qc_ood = ((KL >= quantile(KL, 0.75)) * qc_sampled).mean(0) -- the target cost critics run
        # on the rows that pass only: n - floor(0.75 (n - 1)) - 1 of the n = N*B (torch.quantile 'linear'; ties add rows)
        n = N * Bsz
        kept = n - int(0.75 * (n - 1)) - 1
        fx -= 2.0 * 2 * lin([od + ad] + H + [1]) * (n - kept)
    kc, ke = share
    fx -= 2.0 * (2 * 16 * kc * H[0] + 16 * ke * V) * (N * Bsz) * 4 / 5  # (2 target cost critics; the VAE encoder)
    return fx


def _share(eng):
    """(k16 of the cost-critic launch, k16 of the encoder launch) of an engine's shared-observation tiles, (0, 0) = plain."""
    return int(getattr(eng, "pre_cost", 0) or 0), int(getattr(eng, "pre_enc", 0) or 0)


def lease_diagnostics(device) -> dict:
    """Facts about the box this run got (the driver's fresh lease is not the build's): clocks, power, partition modes,
    runtime / driver versions, the HIP/HSA environment, and where the HIP runtime keeps kernel arguments -- the one that
    explained round 2's 1700-vs-2155 gap (host-resident kernargs, profiles/r3_kernarg_ab.txt)."""
    import ctypes
    import subprocess
    d = {}
    try:
        pr = torch.cuda.get_device_properties(device)
        d["device"] = {"name": pr.name, "arch": getattr(pr, "gcnArchName", None), "cus": pr.multi_processor_count,
                       "hbm_gib": round(pr.total_memory / 2 ** 30, 1),
                       "clock_mhz": round(getattr(pr, "clock_rate", 0) / 1e3)}
    except Exception as e:
        d["device"] = {"error": repr(e)[:120]}
    d["versions"] = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    try:
        d["versions"]["rocm"] = open("/opt/rocm/.info/version").read().strip()
    except Exception:
        pass
    d["env"] = {k: v for k, v in sorted(os.environ.items())
                if k.split("_")[0] in ("HSA", "GPU", "HIP", "ROCR", "AMD", "NCCL", "RCCL", "OSRL", "DEBUG")}
    try:  # device index as rocm-smi counts it: the first visible device
        vis = (os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES") or "").split(",")[0]
        idx = str(int(vis) + (device.index or 0)) if vis.isdigit() else str(device.index or 0)
        out = subprocess.run(["rocm-smi", "-d", idx, "--showclocks", "--showpower", "--showmaxpower", "--showperflevel",
                              "--showcomputepartition", "--showmemorypartition", "--showdriverversion", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        js = json.loads(out[out.index("{"):])
        card = next((v for k, v in js.items() if k.startswith("card")), {})
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "partition", "performance level")):
                keep[k] = v
        keep.update({k: v for k, v in js.get("system", {}).items()})
        d["smi"] = keep
    except Exception as e:
        d["smi"] = {"error": repr(e)[:120]}
    try:
        from osrl_amd import _lib as L
        scratch = torch.zeros(2, dtype=torch.int64, device=device)
        where, addr = ctypes.c_int32(-2), ctypes.c_uint64(0)
        L.check(L.load().osrl_kernarg_probe(scratch.data_ptr(), ctypes.byref(where), ctypes.byref(addr),
                                            torch.cuda.current_stream().cuda_stream), "osrl_kernarg_probe")
        d["kernargs_in"] = {1: "device memory", 0: "HOST memory (every wave fetches launch arguments over PCIe)",
                            -1: "unknown"}.get(where.value, "unknown")
    except Exception as e:
        d["kernargs_in"] = "probe failed: " + repr(e)[:120]
    return d


class Workload:
    """One BASELINE config as ``step()`` = one train step on a minibatch drawn on device from a HBM-resident store."""

    def __init__(self, name: str, device, rank: int, world: int, dp, n_store: int = 1 << 20, seed: int = 0,
                 use_graph: bool = True, steps_per_graph: int = 1):
        from osrl_amd.common.replay import ReplayStore, SequenceStore, synthetic_transitions
        cfg = self.cfg = CONFIGS[name]
        self.name, self.device, self.use_graph = name, device, use_graph
        od, ad, B = cfg["od"], cfg["ad"], cfg["B"]
        dev = str(device)
        torch.manual_seed(seed)
        kw = dict(rows_global=B * world, dist=dp) if dp is not None else {}
        if cfg["algo"] in ("cpq", "bcql"):
            kw_seed = dict(seed=1234, **kw) if dp is not None else {}
        shard = n_store // max(world, 1)
        if cfg["algo"] == "cpq":
            from osrl_amd.algorithms import CPQ, CPQTrainer
            self.model = CPQ(od, ad, 1.0, HID, HID, VAE_H, NS, 0.99, 0.005, 0.5, 2, 2, 1.5, 10, cfg["episode_len"], device=dev)
            self.trainer = CPQTrainer(self.model, None, None, actor_lr=1e-4, critic_lr=1e-3, alpha_lr=1e-4, vae_lr=1e-3,
                                      reward_scale=0.1, cost_scale=1.0, device=dev, stats_mode="none")
            self.eng = self.model.engine(B, **kw_seed)
        elif cfg["algo"] == "bcql":
            from osrl_amd.algorithms import BCQL, BCQLTrainer
            self.model = BCQL(od, ad, 1.0, HID, HID, VAE_H, NS, 0.99, 0.005, 0.05, 0.75, 0.5, [0.1, 0.003, 0.001], 2, 2,
                              10, cfg["episode_len"], device=dev)
            self.trainer = BCQLTrainer(self.model, None, None, 1e-3, 1e-3, 1e-3, stats_mode="none")
            self.eng = self.model.engine(B, **kw_seed)
        elif cfg["algo"] == "bc":
            from osrl_amd.algorithms import BC, BCTrainer
            self.model = BC(od, ad, 1.0, HID, cfg["episode_len"], device=dev)
            self.trainer = BCTrainer(self.model, None, None, actor_lr=1e-3, stats_mode="none")
            self.eng = self.model.engine(B, **kw)
        else:
            from osrl_amd.algorithms import CDT, CDTTrainer
            T = cfg["T"]
            self.model = CDT(od, ad, 1.0, seq_len=T, episode_len=cfg["episode_len"], embedding_dim=cfg["E"],
                             num_layers=cfg["layers"], num_heads=cfg["heads"], attention_dropout=0.1, residual_dropout=0.1,
                             embedding_dropout=0.1, use_rew=True, use_cost=True, cost_transform=True, stochastic=True,
                             target_entropy=-ad, device=dev)
            self.trainer = CDTTrainer(self.model, None, None, learning_rate=1e-4, weight_decay=1e-4, clip_grad=0.25,
                                      lr_warmup_steps=500, loss_cost_weight=0.02, stats_mode="none", seed=1234)
            self.eng = self.model.engine(B, self.trainer.cfg, dist=dp)
        if cfg["algo"] == "cdt":
            rs = np.random.RandomState(1 + rank)
            n_traj, EL = max(64, (shard >> 3) // cfg["episode_len"]), cfg["episode_len"]
            trajs = []
            for _ in range(n_traj):
                c = (rs.uniform(size=EL) < 0.1).astype(np.float32)
                r = rs.uniform(0, 1, EL).astype(np.float32)
                trajs.append(dict(observations=rs.randn(EL, od).astype(np.float32),
                                  actions=rs.uniform(-1, 1, (EL, ad)).astype(np.float32),
                                  returns=np.cumsum(r[::-1])[::-1].copy(), cost_returns=np.cumsum(c[::-1])[::-1].copy(),
                                  costs=c))
            self.store = SequenceStore(trajs, T, device, reward_scale=0.1, cost_scale=1.0, seed=1, rank=rank)
            self.eng.attach_store(self.store)
            self._step = lambda: self.eng.step_store(self.use_graph)
        else:
            self.store = ReplayStore(synthetic_transitions(shard, od, ad, seed=1 + rank), device, reward_scale=0.1,
                                     cost_scale=1.0, seed=1, rank=rank, world=1)
            self.eng.attach_replay(self.store)
            self._step = lambda: self.eng.step_replay(self.use_graph)
        # several steps per hipGraph, software-pipelined across steps (osrl_amd/engine/pipeline.py): CPQ / BCQ-Lag on one
        # GPU.  Both graphs (n-step and one-step) are captured HERE, outside any timed region
        self.pipe = None
        if steps_per_graph != 1:
            self.build_pipe(steps_per_graph)

    def build_pipe(self, steps_per_graph: int = 0) -> int:
        """Several steps per replayed graph (0 = the plan's choice); both graphs -- n-step and one-step -- are captured
        here, outside any timed region.  Returns the steps per graph in effect."""
        cfg, eng = self.cfg, self.eng
        if not (self.use_graph and getattr(eng, "dist", None) is None and cfg["algo"] in ("cpq", "bcql")):
            return 1
        spg = int(steps_per_graph) or int(eng.plan.steps_per_graph)
        if spg > 1:
            from osrl_amd.engine.pipeline import PipelinedSteps
            self.pipe = eng._pipe = PipelinedSteps(eng, spg)
            self.pipe.capture()
            eng.capture()
        return spg

    def step(self) -> None:
        self._step()

    def run(self, n: int) -> None:
        """EXACTLY n train steps: whole pipelined graphs + the remainder as single-step replays (or n single steps)."""
        if self.pipe is not None:
            self.pipe.run(n)
        else:
            for _ in range(n):
                self._step()

    def api_batch(self):
        """Device-resident synthetic batch for the Trainer-API path (train_one_step(tensors))."""
        cfg, dev = self.cfg, self.device
        B, od, ad = cfg["B"], cfg["od"], cfg["ad"]
        g = torch.Generator(device="cpu").manual_seed(5)
        f = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
        u = lambda *s: torch.rand(*s, generator=g).to(dev)  # noqa: E731
        if cfg["algo"] == "bc":
            return (f(B, od), (u(B, ad) * 2 - 1))
        if cfg["algo"] == "cdt":
            T = cfg["T"]
            start = torch.randint(0, cfg["episode_len"], (B, 1), generator=g).to(dev)
            mask = torch.ones(B, T, device=dev)
            mask[::10, T - 5:] = 0
            return (f(B, T, od), u(B, T, ad) * 2 - 1, u(B, T) * 10, u(B, T) * 20,
                    start + torch.arange(T, device=dev)[None], mask, u(B) * 20, (u(B, T) < 0.1).float())
        return (f(B, od), f(B, od), u(B, ad) * 2 - 1, f(B), (u(B) < 0.1).float(), (u(B) < 0.01).float())


def timed_steps(step, steps: int, warmup: int, barrier=None):
    """``step``: a callable for ONE step, or a Workload (its ``run(n)`` = exactly n steps, possibly several per graph)."""
    import gc
    run = step.run if hasattr(step, "run") else (lambda n: [step() for _ in range(n)])
    run(warmup)
    torch.cuda.synchronize()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    # the interpreter's cyclic collector stays out of the timed region (as `timeit` does).  NOT collected here: a
    # gc.collect() in front of the region is a > 20 ms idle gap for the device, after which the first replays run 5 % slow
    # (the power-state ramp of preroll(); measured: 2085-2110 vs 2210-2225 steps/s, gpurun_out/r6p)
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        if barrier:
            barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        if was_enabled:
            gc.enable()
    return dt


def preroll(device, ms: float = 40.0) -> float:
    """Dense fp32 GEMM work queued right in front of the warm-up steps, nothing synchronising in between.

    After ANY idle gap (20 ms is enough: tools/step_series.py, profiles/r3_step_series.txt) the device runs the first
    ~25 replays of the CPQ step at 470-485 us and only then settles at 455 us -- a power-state ramp, not a property of the
    step: behind 40 GEMMs the very first replays run at 450.  The driver's command times steps 6..25 of the process
    (W = 5, K = 20 = 9 ms), i.e. exactly that ramp: 2105-2152 steps/s where the same 20 steps after 200 warm-up steps
    give 2208-2233 and 200 steps give 2202-2207 (profiles/r3_warmup_ab.txt).  The pre-roll is untimed, touches no engine
    state and issues no step; it makes the K timed steps measure the rate a training job runs at from its 30th millisecond
    on.  ``--preroll-ms 0`` turns it off.  Returns the milliseconds queued."""
    if ms <= 0:
        return 0.0
    x = torch.randn(4096, 4096, device=device)
    y = torch.empty_like(x)
    torch.mm(x, x, out=y)  # (first call: library setup)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.mm(x, x, out=y)
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1), 0.05)
    n = max(1, int(ms / one + 0.5))
    for _ in range(n):
        torch.mm(x, x, out=y)
    return n * one


def time_kernel(fn, iters=30):
    """Average duration (s) of ``fn`` (one kernel launch on the current stream) by HIP events."""
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def mlp_fwd_flops(run):
    d = run.net.dims
    return 2.0 * run.rows * run.net.E * lin(d)


DEFAULT_STEPS_PER_GRAPH = 0  # 0 = what the engine's plan says (engine/plan.py steps_per_graph: C2 20, C3 10, C4 8; the CPQ graphs not joined inside)

PROBE_SITES = ("enc_ood", "costold_ood", "vae_dw", "actor_phase_fwd", "critic_fwd")


def in_step_us(eng, iters=40):
    """Durations of the step's five heaviest launches AS THEY RUN INSIDE THE STEP: the step body is issued eagerly on
    the same two streams the captured graph uses (main + side branch), with HIP events recorded on the launching stream
    right around each of them, so whatever the other branch runs beside a launch runs beside it here too.  Returns
    {site: (mean us, median us)}; "enc_ood" is the dominant one.  The rocprofv3 --kernel-trace --stats summary of this
    command (profiles/) lists the same kernels' averages over graph replays."""
    from osrl_amd.engine.core import Branches
    par = Branches(True, 2 if eng.dist is not None else 1)
    mk = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))  # noqa: E731
    evs = [{k: mk() for k in PROBE_SITES} for _ in range(iters)]
    snap = eng._snapshot()
    try:
        for i in range(3 + iters):
            eng._probe = evs[i - 3] if i >= 3 else None
            eng.body(True, par)
        torch.cuda.synchronize()
        res = {}
        for k in PROBE_SITES:
            ts = sorted(e[k][0].elapsed_time(e[k][1]) * 1e3 for e in evs)
            res[k] = (float(np.mean(ts)), float(ts[len(ts) // 2]))
    finally:
        eng._probe = None
        torch.cuda.synchronize()
        eng._restore(snap)
    return res


def in_graph_us(eng, sites=("enc_ood", "costold_ood"), iters=60):
    """Durations of named launches INSIDE THE REPLAYED GRAPH: the step is captured with a one-lane stamp launch
    (osrl_stamp_realtime: the device's 100 MHz real-time counter) in front of and behind each named launch, on that
    launch's own stream, plus one more right behind the closing stamp (stamp-to-stamp = one queue boundary, subtracted
    twice: the bracket holds two); after each replay the stamps are read.  This is the figure the rocprofv3 --kernel-trace
    average of the same kernel (profiles/) has to agree with -- HIP events around EAGERLY issued launches (in_step_us)
    do not: the host cannot keep two queues fed the way a replayed graph does, so the overlap differs.  The stamped
    graph is thrown away afterwards (the timed region captures a clean one).  Returns {site: (mean us, median us)}."""
    buf = torch.zeros(3 * len(sites), dtype=torch.int64, device=eng.dev)
    snap = eng._snapshot()
    res = {}
    try:
        eng._probe = {k: buf.data_ptr() + 24 * i for i, k in enumerate(sites)}
        eng.graph = None
        eng.capture()
        samples = {k: [] for k in sites}  # (+ "<site>/boundary": the stamp-to-stamp cost that was subtracted twice)
        for it in range(5 + iters):
            eng.graph.replay()
            eng.st.host_step += 1
            torch.cuda.synchronize()
            if it >= 5:
                v = buf.tolist()
                for i, k in enumerate(sites):
                    # (t1 - t0) = the launch + TWO queue boundaries (stamp -> launch, launch -> stamp); the adjacent pair
                    # (t2 - t1) is one such boundary measured in place; 10 ns ticks -> us
                    pair = (v[3 * i + 2] - v[3 * i + 1]) * 0.01
                    samples[k].append((v[3 * i + 1] - v[3 * i]) * 0.01 - 2.0 * pair)
                    samples.setdefault(k + "/boundary", []).append(pair)
        for k, ts in samples.items():
            ts.sort()
            res[k] = (float(np.mean(ts)), float(ts[len(ts) // 2]))
    finally:
        eng._probe = None
        eng.graph = None
        torch.cuda.synchronize()
        eng._restore(snap)
    return res


def collectives_in_step(eng, dp, iters=30):
    """Every collective of the data-parallel step AS IT RUNS INSIDE THE STEP: the step body is issued eagerly on the
    graph's two streams (as in_step_us does) with the DataParallel hook recording HIP events around each collective on
    its stream.  Every rank runs this (the bodies hold the collectives); returns [{what, bytes, us, us_median}] in issue
    order -- 4 entries for CPQ: VAE gradient, [critic | cost-critic] gradients, KL all-gather, [actor | statistics |
    qc_ood] -- whose sum against ms_per_step is the step's exposed communication."""
    from osrl_amd.engine.core import Branches
    par = Branches(True, 2 if eng.dist is not None else 1)
    snap = eng._snapshot()
    recs = []
    try:
        for i in range(3 + iters):
            dp._probe = [] if i >= 3 else None
            eng.body(True, par)
            if i >= 3:
                recs.append(dp._probe)
        torch.cuda.synchronize()
        n = min(len(r) for r in recs)
        out = []
        for j in range(n):
            ts = sorted(r[j][2].elapsed_time(r[j][3]) * 1e3 for r in recs)
            out.append({"what": recs[0][j][0], "bytes": recs[0][j][1], "us": round(float(np.mean(ts)), 2),
                        "us_median": round(float(ts[len(ts) // 2]), 2)})
    finally:
        dp._probe = None
        torch.cuda.synchronize()
        eng._restore(snap)
    return out


def roofline(eng, cfg_name="c2"):
    """Dominant kernels of the CPQ step: the two N*B-row forward launches (half of the step's issued FLOPs).  Needs no
    trained state (the in-step probe snapshots and restores the engine), so main() runs it BEFORE the timed region;
    ``step_frac`` is filled in afterwards.  The headline entry is the launch that takes the most time INSIDE the step (=
    the top kernel of the rocprofv3 --kernel-trace --stats summary of this command under profiles/); both launches are
    carried under ``kernels`` with their in-step and isolated figures, algorithmic bytes and counted HBM traffic."""
    from osrl_amd import _lib as L
    cands = {
        # (the launches exactly as the step issues them: on tiles of shared observations where the plan says so, plan.ood_share)
        "mlp_fwd<vae-encoder, N*B rows>": (eng.r_enc_ood, lambda: eng.r_enc_ood.forward(
            eng.obs, eng.sampled, map0=L.MAP_MOD, div0=eng.B, share_k16=_share(eng)[1]), "enc_ood",
            "mlp_fwd_nb8_pre_kernel_p" if _share(eng)[1] else "mlp_fwd_nb8_kernel_p", _share(eng)[1]),
        "mlp_fwd<cost_critic_old x2, N*B rows>": (eng.r_costold_ood, lambda: eng.r_costold_ood.forward(
            eng.obs, eng.sampled, map0=L.MAP_MOD, div0=eng.B, share_k16=_share(eng)[0]), "costold_ood",
            "mlp_fwd_nb_kernel_p<4, false, false, true>" if _share(eng)[0] else "mlp_fwd_nb_kernel_p<4, false>", _share(eng)[0]),
    }
    # the in-step probe runs whole step bodies: under data parallelism those contain collectives, which rank 0 must not
    # issue out of step with its peers -- N > 1 reports the isolated figure only.  It goes first: its
    # step bodies leave a sampled minibatch and the N*B sampled actions in the buffers the isolated launches read
    sites = in_step_us(eng) if eng.dist is None else {}
    # ... and the two N*B-row launches as they run inside the REPLAYED graph (device-side stamps): the headline figure
    try:
        graph_sites = in_graph_us(eng) if eng.dist is None and eng.replay is not None else {}
    except Exception as e:  # (a failing probe must not take the line down: the eager figure is reported instead)
        graph_sites = {"error": repr(e)[:200]}
    if eng.dist is not None:  # no step has run yet: time the launches on data, not on the zero-initialised buffers
        eng.obs.normal_()
        eng.sampled.normal_()
    # HBM bytes per launch come from hardware counters, which cannot be read from inside this process: the figures are the
    # ones a separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass pair measured for THESE launches of THIS
    # config (tools/pmc_nb.py under tools/gpu_r5_pmc.sh -> profiles/pmc_traffic.json, keyed by config)
    pmc, traffic_src = {}, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            js = json.load(open(pmc_path))
            pmc = js.get(cfg_name, {}) if isinstance(js.get(cfg_name), dict) else {}
            traffic_src = js.get("source", "static: profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / "
                                           "WRITE_SIZE passes of tools/gpu_r5_pmc.sh; not measured in this run)")
        except Exception:
            pmc = {}
    res = {}
    for name, (run, fn, site, sym, k16) in cands.items():
        t = time_kernel(fn)
        fl_alg = mlp_fwd_flops(run)
        # FLOPs the launch ISSUES: on tiles of shared observations the first 16 k16 input columns of layer 0 are multiplied for
        # one of a tile's five row blocks (csrc/mlp_nb.hip nb_share_acc) -- the fraction is priced on these, not on fl_alg
        fl = fl_alg - 2.0 * run.net.E * 16 * k16 * run.net.dims[1] * run.rows * 4 / 5
        mean_us, med_us = graph_sites.get(site, (float("nan"), float("nan"))) if "error" not in graph_sites else (float("nan"),) * 2
        how = "graph"
        if mean_us != mean_us:  # no in-graph figure: the eager two-stream probe
            mean_us, med_us = sites.get(site, (float("nan"), float("nan")))
            how = "eager"
        in_run = mean_us == mean_us
        fl_in = fl
        if site == "costold_ood" and getattr(eng, "ood_rows", False):
            # plan.ood_rows: inside the step this launch runs on the selected rows only (the isolated figure below is the
            # same kernel form on all N*B rows); its in-step FLOPs are those of the rows the last probe step selected
            fl_in = fl * float(int(eng.ood_count[0].item())) / float(run.rows)
        d = run.net.dims
        # algorithmic bytes of the launch as SURVEY.md 8d counts them: every row of the virtual [N*B, in] input, the weights
        # + biases of every net once, the [rows, out] result.  `distinct_bytes`: the same with the input's DISTINCT rows only
        # (row r reads observation r % B and sampled action r: B x obs_dim + N*B x act_dim floats) -- the counted traffic
        # lies between the two (the 80-row tiles re-read observation rows that other tiles already brought on die)
        wts = run.net.E * (lin(d) + sum(d[1:]))
        alg = 4 * (run.rows * d[0] + wts + run.net.E * run.rows * d[-1])
        distinct = 4 * (eng.B * eng.obs.shape[1] + run.rows * eng.sampled.shape[1] + wts + run.net.E * run.rows * d[-1])
        res[name] = dict(symbol=sym, gflop=round(fl / 1e9, 3), gflop_plain_tiles=round(fl_alg / 1e9, 3), share_k16=int(k16),
                         isolated_us=round(t * 1e6, 2),
                         isolated_frac=round(fl / t / 1e12 / PEAK_FP32_TFLOPS, 4),
                         in_step_us=round(mean_us, 2) if in_run else None,
                         in_step_us_median=round(med_us, 2) if in_run else None,
                         in_step_frac=round(fl_in / (mean_us * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4) if in_run else None,
                         in_step_gflop=round(fl_in / 1e9, 3),
                         in_step_how=how if in_run else None,
                         stamp_boundary_us=round(graph_sites[site + "/boundary"][0], 2) if site + "/boundary" in graph_sites else None, in_step_eager_us=round(sites[site][0], 2) if site in sites else None,
                         algorithmic_bytes=int(alg), distinct_bytes=int(distinct), traffic=pmc.get(name), wg_cap=int(run.fwd_c.wg_cap), _fl=fl, _t=t,
                         _us=mean_us, _fl_in=fl_in)
    have_run = all(v["in_step_us"] is not None for v in res.values())
    # dominant = the launch that takes the most time inside the step (N > 1: the most FLOPs)
    dom = max(res, key=(lambda k: res[k]["_us"]) if have_run else (lambda k: res[k]["_fl"]))
    r = res[dom]
    ach_iso = r["_fl"] / r["_t"] / 1e12
    ach = r["_fl_in"] / (r["_us"] * 1e-6) / 1e12 if have_run else ach_iso
    out = {"bound": "mfma", "kernel": dom, "symbol": r["symbol"], "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS,
           "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
           "frac_is": ("in-run: the launch inside the REPLAYED one-step graph, bracketed by device-side 100 MHz stamps on its own "
                       "stream (in_graph_us; cross-check: the rocprofv3 --kernel-trace average of the same kernel over the one-step "
                       "graph, 106.8 us in profiles/r6c_bench_kernel_stats_c2_1step.csv; over the 20-step graph the timed region replays "
                       "the profiler's average is 99.9 us, r6c_bench_kernel_stats_c2.csv, and the un-profiled start stamps of that "
                       "graph, r6c_trace_unprofiled_c2.txt, bound it by <= 92 us: how the two queues line up differs between the "
                       "graphs and under rocprofv3).  'Dominant' is PER LAUNCH: kernel = the single launch with the largest in-step duration; by TOTAL "
                       "time per step the top symbol is another one (top_by_total)" if r.get("in_step_how") == "graph" else
                       "in-run (eagerly issued two-stream step body, HIP events)") if have_run else
                      "isolated (N > 1: no in-step probe)",
           # the symbol with the largest TOTAL time per step in the rocprofv3 --kernel-trace --stats summary of this command
           # (profiles/r6c_bench_kernel_stats_c2.csv): three launches per step of the 16-row paired forward (actor trunks,
           # critic-phase and cost-phase forwards), ~127 us per step between them at MFMA-busy 0.28 (profiles/r6_pmc_c2.json)
           "top_by_total": {"symbol": "mlp_fwd2_kernel_p<1, 2, 8>", "launches_per_step": 3,
                            "source": "static: profiles/r6c_bench_kernel_stats_c2.csv"} if cfg_name == "c2" else None,
           "isolated_achieved": round(ach_iso, 3), "isolated_frac": round(ach_iso / PEAK_FP32_TFLOPS, 4),
           # what a loop of nothing but independent v_mfma_f32_16x16x4_f32 sustains on all 256 CUs (one / two waves per SIMD):
           # the nominal peak above assumes 32 cycles per instruction, the chip delivers 37-41.  Reported beside `peak`, never
           # instead of it.
           "sustained_peak": {"value": [122.1, 134.6], "unit": "TFLOP/s", "frac_of_it": round(ach / 134.6, 4),
                              "isolated_frac_of_it": round(ach_iso / 134.6, 4),
                              "source": "static: tools/mfma_bf16_probe.hip, profiles/r6_mfma_bf16_probe.txt"},
           "traffic": r["traffic"], "traffic_source": traffic_src if r["traffic"] is not None else None,
           "algorithmic_bytes": r["algorithmic_bytes"],
           # five launches of the step by HIP events around EAGERLY issued launches (two streams): indicative only
           "in_step_sites_us": {k: round(v[0], 2) for k, v in sites.items()} or None,
           "in_step_sites_how": "eager two-stream step body, HIP events (not the replayed graph)" if sites else None,
           "isolated_us": r["isolated_us"], "in_step_us": r["in_step_us"], "in_step_us_median": r["in_step_us_median"],
           "in_step_frac": r["in_step_frac"],
           "step_frac": None,  # main(): algorithmic GFLOP per step / measured ms per step / peak
           "kernels": {k: {kk: vv for kk, vv in v.items() if not kk.startswith("_")} for k, v in res.items()}}
    return out


def cpu_baseline(budget_s=24.0):
    """The CPU side of the same workload (CPQ (76,2) B=2048), bounded to ~budget_s of CPU work, two restatements:

      * oracle/torch_cpq_cpu.py -- torch CPU tensors + autograd + torch.optim.Adam: what the reference's own CPU path is
        made of (train_cpq.py:35 pins 4 threads); probed at 4 threads and at wider pools;
      * oracle/osrl_oracle.py   -- the numpy port with the hand-derived backward (the parity checker).

    ``value`` is the FASTEST of them at its best thread count (the strongest CPU baseline this host gives); the others
    are listed in ``variants``.  A reported baseline, not a target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import Case, hyper, make_batch, make_noise, make_params
    from oracle.torch_cpq_cpu import TorchCPQ
    from oracle_util import build_oracle
    cfg = CONFIGS["c2"]
    c = Case("bench_c2", "cpq", od=cfg["od"], ad=cfg["ad"], B=cfg["B"], hidden=HID, vae_hidden=VAE_H, N=NS, steps=1,
             episode_len=1000)
    hp = hyper(c)
    o = build_oracle(c)
    tc = TorchCPQ(make_params(c), max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"], tau=hp["tau"],
                  beta=hp["beta"], qc_scalar=hp["qc_scalar"], cost_limit=c.cost_limit, episode_len=c.episode_len,
                  actor_lr=hp["actor_lr"], critic_lr=hp["critic_lr"], alpha_lr=hp["alpha_lr"], vae_lr=hp["vae_lr"])
    b, nz = make_batch(c), make_noise(c, 0)
    args = (b["observations"], b["next_observations"], b["actions"], b["rewards"], b["costs"], b["done"], nz)
    ncpu = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None

    def loop(step, budget, max_steps=400):
        step(*args)
        t0 = time.perf_counter()
        n = 0
        while True:
            step(*args)
            n += 1
            if time.perf_counter() - t0 > budget or n >= max_steps:
                break
        return n, time.perf_counter() - t0

    def run_numpy(nthreads, budget):
        ctx = threadpool_limits(limits=nthreads, user_api="blas") if threadpool_limits else None
        try:
            return loop(o.train_one_step, budget)
        finally:
            if ctx is not None:
                ctx.unregister() if hasattr(ctx, "unregister") else ctx.__exit__(None, None, None)

    def run_torch(nthreads, budget):
        keep = torch.get_num_threads()
        torch.set_num_threads(nthreads)
        try:
            return loop(tc.train_one_step, budget)
        finally:
            torch.set_num_threads(keep)

    # both thread pools oversubscribe on big hosts: probe a few widths briefly, keep the fastest
    cands = sorted({k for k in (4, 8, 16, 32, 64) if k <= ncpu} | ({ncpu} if ncpu <= 64 else set()))
    np_cands = cands if threadpool_limits else [ncpu]
    slice_s = budget_s / (2.0 * (len(cands) + len(np_cands)))
    rate = lambda r: r[0] / r[1]  # noqa: E731
    probe = {("torch", k): run_torch(k, slice_s) for k in cands}
    probe.update({("numpy", k): run_numpy(k, slice_s) for k in np_cands})
    (impl, best) = max(probe, key=lambda k: rate(probe[k]))
    n, dt = (run_torch if impl == "torch" else run_numpy)(best, budget_s / 2.0)
    variants = {f"{i}@{k}": round(rate(r), 2) for (i, k), r in sorted(probe.items())}
    return {"value": round(n / dt, 3), "unit": "grad-steps/s", "cores": int(best), "kind": "port",
            "implementation": {"torch": "oracle/torch_cpq_cpu.py (torch CPU + autograd + torch.optim.Adam)",
                               "numpy": "oracle/osrl_oracle.py (numpy, hand-derived backward)"}[impl],
            "variants": variants,
            "sample": f"{n} CPQ steps (76,2) B=2048 in {dt:.1f}s with the {impl} restatement at {best} threads -- the "
                      f"fastest of {len(probe)} (implementation, threads) pairs probed for {slice_s:.1f}s each on a "
                      f"{ncpu}-cpu host; the reference's default (torch, 4 threads): "
                      f"{rate(probe[('torch', min(cands))]):.2f} steps/s"}


def cpu_baseline_others(which=("c1", "c3", "c4", "c5")):
    """CPU figures for the other BASELINE configs (SURVEY.md 8d's timing plan: "the build's own plain-PyTorch restatement"
    on the host cores): the torch CPU + autograd + torch.optim restatement of each algorithm (oracle/torch_cpu_baselines.py
    for BC / BCQ-Lag / CDT, oracle/torch_cpq_cpu.py for C4's CPQ; each pinned to the reference's golden vectors by
    tests/test_oracle_golden.py) on the config's shapes, at 4 threads (the reference's default, *_configs.py ``threads``)
    and at a wide pool, plus the numpy oracle at 4 BLAS threads.  Bounded samples: c1 ~1 s per setting; c3 / c4 >= 3
    steps; c5 on a 128-sample slice of the 1024-sample batch with dropout 0.1 active (the loss is a sum over samples; the
    figure is divided by 8) -- reported baselines beside the GPU numbers, not targets."""
    import dataclasses
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import Case, hyper, make_batch, make_cdt_batch, make_cdt_params, make_noise, make_params
    from oracle.torch_cpq_cpu import TorchCPQ
    from oracle.torch_cpu_baselines import TorchBC, TorchBCQL, TorchCDT
    from oracle_util import build_oracle
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    ncpu = os.cpu_count() or 1
    wide = min(ncpu, 16)  # C2's probe (cpu_baseline) finds the torch optimum at 8-16 threads on the 256-cpu hosts

    def at_blas(nthreads, fn):
        ctx = threadpool_limits(limits=nthreads, user_api="blas") if threadpool_limits else None
        try:
            return fn()
        finally:
            if ctx is not None:
                ctx.unregister() if hasattr(ctx, "unregister") else ctx.__exit__(None, None, None)

    def at_torch(nthreads, fn):
        keep = torch.get_num_threads()
        torch.set_num_threads(nthreads)
        try:
            return fn()
        finally:
            torch.set_num_threads(keep)

    def rate(step, min_steps, budget):
        step()
        t0, n = time.perf_counter(), 0
        while n < min_steps or time.perf_counter() - t0 < budget:
            step()
            n += 1
            if n >= 2000:
                break
        return n, time.perf_counter() - t0

    # the first seconds of BLAS work in a process run 10-50x slow on some hosts (thread-pool spin-up / cpu quota ramp:
    # 16 vs 970 BC steps/s measured back to back in this container): 3 s of matmuls first
    wa = np.random.RandomState(0).randn(512, 512).astype(np.float32)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        wa @ wa
    out = {}
    for name in which:
        try:
            cfg = CONFIGS[name]
            np_step = None
            if cfg["algo"] == "cdt":
                from test_gpu_cdt import C5_FULL
                from test_oracle_cdt_golden import build_cdt_oracle
                c = dataclasses.replace(C5_FULL, B=128)
                t = TorchCDT(make_cdt_params(c), seq_len=c.T, num_heads=c.heads, num_layers=c.layers,
                             cost_transform=c.cost_transform, stochastic=c.stochastic, init_temperature=0.1,
                             target_entropy=-c.ad, learning_rate=c.lr, weight_decay=c.wd, clip_grad=c.clip,
                             lr_warmup_steps=c.warmup, loss_cost_weight=c.cost_w, loss_state_weight=c.state_w, dropout=0.1)
                b = make_cdt_batch(c)
                a = (b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                     b["episode_cost"], b["costs"])
                step, scale, mins, budget = (lambda: t.train_one_step(*a)), 128.0 / cfg["B"], 2, 0.0
                o = build_cdt_oracle(dataclasses.replace(c, dropout=0.0), np.float32)
                np_step = lambda: o.train_one_step(*a)  # noqa: E731
                what = (f"torch restatement of CDT (oracle/torch_cpu_baselines.py TorchCDT, dropout 0.1) on a 128-sample slice "
                        f"of the {cfg['B']}-sample batch, x 1/8")
            else:
                c = Case("bench_" + name, cfg["algo"], od=cfg["od"], ad=cfg["ad"], B=cfg["B"], hidden=HID, vae_hidden=VAE_H,
                         N=NS, steps=1, episode_len=cfg["episode_len"])
                hp, b, o = hyper(c), make_batch(c), build_oracle(c)
                scale = 1.0
                if cfg["algo"] == "bc":
                    t = TorchBC(make_params(c), c.max_action, hp["actor_lr"])
                    step, mins, budget = (lambda: t.train_one_step(b["observations"], b["actions"])), 20, 1.0
                    np_step = lambda: o.train_one_step(b["observations"], b["actions"])  # noqa: E731
                else:
                    nz = make_noise(c, 0)
                    args_ = (b["observations"], b["next_observations"], b["actions"], b["rewards"], b["costs"], b["done"], nz)
                    if cfg["algo"] == "cpq":
                        t = TorchCPQ(make_params(c), max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"],
                                     tau=hp["tau"], beta=hp["beta"], qc_scalar=hp["qc_scalar"], cost_limit=c.cost_limit,
                                     episode_len=c.episode_len, actor_lr=hp["actor_lr"], critic_lr=hp["critic_lr"],
                                     alpha_lr=hp["alpha_lr"], vae_lr=hp["vae_lr"])
                    else:
                        t = TorchBCQL(make_params(c), max_action=c.max_action, sample_action_num=c.N, gamma=hp["gamma"],
                                      tau=hp["tau"], phi=hp["phi"], lmbda=hp["lmbda"], beta=hp["beta"], PID_gains=hp["PID"],
                                      cost_limit=c.cost_limit, episode_len=c.episode_len, actor_lr=hp["actor_lr"],
                                      critic_lr=hp["critic_lr"], vae_lr=hp["vae_lr"])
                    step, mins, budget = (lambda: t.train_one_step(*args_)), 3, 1.0
                    np_step = lambda: o.train_one_step(*args_)  # noqa: E731
                what = (f"torch restatement of {cfg['algo']} (oracle/torch_c{'pq_cpu' if cfg['algo'] == 'cpq' else 'pu_baselines'}.py) "
                        f"at ({cfg['od']}, {cfg['ad']}) B={cfg['B']}")
            n4, d4 = at_torch(min(4, ncpu), lambda: rate(step, mins, budget))
            nw, dw = at_torch(wide, lambda: rate(step, mins, budget)) if wide > 4 else (n4, d4)
            r4, rw = n4 / d4 * scale, nw / dw * scale
            variants = {"torch@4": round(r4, 4), f"torch@{wide}": round(rw, 4)}
            if np_step is not None:
                nn_, dn = at_blas(min(4, ncpu), lambda: rate(np_step, 2 if cfg["algo"] != "bc" else 20, 0.0))
                variants["numpy_oracle@4"] = round(nn_ / dn * scale, 4)
            best_threads = wide if rw >= r4 else min(4, ncpu)
            out[name] = {"value": round(max(r4, rw), 4), "unit": "grad-steps/s", "cores": int(best_threads), "kind": "port",
                         "variants": variants,
                         "sample": f"{what}: {n4} steps in {d4:.1f}s at 4 threads (the reference's default), {nw} in "
                                   f"{dw:.1f}s at {wide} (host: {ncpu} cpus)"}
        except Exception as e:  # a failing side measurement must not take the headline line down
            out[name] = {"error": repr(e)[:200]}
    return out


def cost_return_gap(device) -> dict:
    """BASELINE.json's metric, second half: "(+ cost-return gap vs ref)".  After the timed region: the small CPQ / BCQ-Lag /
    BC cases of tests/cases.py are TRAINED through the HIP path (case.steps = 10 gradient steps on the seeded batch with
    the seeded noise -- the steps the train-step goldens pin) and evaluated with the batched on-device ``evaluate()`` on
    the synthetic safe env; the reference side is tests/golden/eval_rollouts_trained.npz = the REFERENCE models trained
    by the reference's own train_one_step on the same inputs and rolled out by the reference's own rollout()
    (cpq.py:294-347; generated by tests/golden/make_golden_eval_trained.py, data only).  Reported per algorithm: mean
    episode return / cost on both sides and the worst per-episode gap over the 12 seeded episodes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import CASES
    from gpu_util import build_gpu, gpu_batch, gpu_step
    from osrl_amd.common.synthetic_env import SyntheticSafeEnv, VecSyntheticSafeEnv
    from osrl_amd.engine.rollout import BatchedRollout
    g = np.load(os.path.join(ROOT, "tests", "golden", "eval_rollouts_trained.npz"), allow_pickle=False)
    E, EL, cost_scale = 12, 25, 2.0  # tests/golden/make_golden_eval.py EVAL
    out = {"what": "per-episode |gpu - reference| after case.steps train steps on both sides; 12 seeded episodes x 25 "
                   "steps of the synthetic safe env; reference = tests/golden/eval_rollouts_trained.npz"}
    for name in ("cpq_small", "bcql_small", "bc_small"):
        try:
            c = CASES[name]
            m, tr, _ = build_gpu(c, device=str(device), use_graph=True)
            b = gpu_batch(c, device=str(device))
            for s_ in range(c.steps):
                gpu_step(tr, c, b, s_)
            m.episode_len = EL
            cs = cost_scale if c.algo != "bc" else 1.0
            tr.cost_scale = cs
            env = SyntheticSafeEnv(c.od, c.ad, 50, seed=1, init_noise=0.7)
            venv = VecSyntheticSafeEnv(env, E, device, base_seed=100)
            tr.env = venv
            if c.algo == "bcql":
                rets, costs, lens = BatchedRollout(m, venv, "bcql", cs, z=torch.tensor(g[name + "_z"], device=device)).run()
            else:
                tr.evaluate(E)
                rets, costs, lens = tr._rollout[1].run()
            ref = g[name]
            out[name] = {"train_steps": int(c.steps),
                         "return_gpu": round(float(np.mean(rets)), 5), "return_ref": round(float(ref[:, 0].mean()), 5),
                         "cost_gpu": round(float(np.mean(costs)), 5), "cost_ref": round(float(ref[:, 1].mean()), 5),
                         "return_gap_max_rel": float(f"{np.max(np.abs(rets - ref[:, 0]) / np.maximum(1.0, np.abs(ref[:, 0]))):.3e}"),
                         "cost_gap_max": round(float(np.max(np.abs(costs - ref[:, 1]))), 5),
                         "lengths_equal": bool(np.array_equal(lens, ref[:, 2]))}
        except Exception as e:  # a failing side measurement must not take the headline line down
            out[name] = {"error": repr(e)[:200]}
    return out


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--preroll-ms", type=float, default=40.0,
                    help="untimed dense GEMM work queued in front of the warm-up steps (power-state ramp; 0 = none)")
    ap.add_argument("--no-cold", action="store_true", help="skip the first (no pre-roll) timed region")
    ap.add_argument("--extras-timeout", type=float, default=240.0,
                    help="N > 1: seconds the side measurements behind the timed region may take before rank 0 prints the "
                         "headline fields alone (0 = no watchdog)")
    ap.add_argument("--no-extras", action="store_true", help="skip api_path / other_configs (N=1 extras)")
    ap.add_argument("--eager", action="store_true", help="no hipGraph (debug)")
    ap.add_argument("--steps-per-graph", type=int, default=DEFAULT_STEPS_PER_GRAPH,
                    help="train steps per replayed hipGraph, software-pipelined across steps (CPQ / BCQ-Lag, one GPU; 0 = the "
                         "plan's choice, 1 = one step per graph).  K timed steps = K // n graphs + K %% n single-step replays")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the N-rank job (one process per GPU over RCCL)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, cmd)
    # stdout carries exactly ONE line, the JSON: everything else any library writes to fd 1 (RCCL prints a version
    # banner through C stdio, flushed only at exit, i.e. AFTER a Python-level print) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # OSRL_DP_EXCHANGE=ipc: the step's exchanges through IPC-mapped device buffers (engine/dist_ipc.py) instead of RCCL
    # launches; OSRL_IPC_ONE_GPU=1 (lab, implies ipc): every rank on device 0 with gloo as the control plane -- a real peer
    # PROCESS on a one-GPU box (the ranks share the device, so `value` is no scaling figure there)
    one_gpu = os.environ.get("OSRL_IPC_ONE_GPU") == "1"
    exchange = "ipc" if one_gpu else os.environ.get("OSRL_DP_EXCHANGE", "rccl")
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dp, rccl_ranks = None, 0
    force_dp = world == 1 and os.environ.get("OSRL_FORCE_DP") == "1"  # debug: the data-parallel step on one rank
    if world > 1 or force_dp:
        import torch.distributed as dist
        backend = "gloo" if one_gpu else "nccl"
        kw_pg = {} if one_gpu else dict(device_id=device)
        if force_dp:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group(backend, rank=0, world_size=1, **kw_pg)
        else:
            dist.init_process_group(backend, **kw_pg)
        if exchange == "ipc":
            from osrl_amd.engine.dist_ipc import IpcDataParallel
            dp = IpcDataParallel(device=device)
        else:
            from osrl_amd.engine.dist import DataParallel
            dp = DataParallel()
        rccl_ranks = dist.get_world_size() if exchange != "ipc" else 0

    wl = Workload(args.config, device, rank, world, dp, use_graph=not args.eager,
                  steps_per_graph=1)  # (the probes below run on the one-step engine; the pipeline is built behind them)
    cfg, eng = wl.cfg, wl.eng

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # The roofline probes go first: they need nothing from the timed steps, and the device has then been busy for ~60 ms
    # when the warm-up steps start -- a timed region as short as the driver's (K = 20 steps = 10 ms after W = 5) measured
    # 4.5 % below a long one on the same box (2059 vs 2155 steps/s, profiles/r2_bench_driver_cmd.json vs r2_bench.json:
    # idle clocks / first replays suspected).  Every rank runs them (no collectives inside: under data parallelism only
    # the isolated launches are timed), rank 0 reports.
    lease = lease_diagnostics(device) if rank == 0 else None
    roof = None
    if not args.no_roofline and cfg["algo"] == "cpq":
        try:
            roof = roofline(eng, args.config)
        except Exception as e:  # a failing probe must not take the headline line down
            roof = {"error": repr(e)[:300]}
            torch.cuda.synchronize()
    def max_over_ranks(x):
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([x], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # Both protocols are measured and reported: first W warm-up + K timed steps straight after the probes
    # (`no_preroll`: what the same command measured in rounds 1-3's records), then the pre-roll + W + K again: `value`.
    spg = wl.build_pipe(args.steps_per_graph) if dp is None else 1  # (graphs captured here, outside the timed regions)
    n_done = 0
    # Graph priming (untimed, counted in the step total): the FIRST time several replays are in flight at once the runtime
    # sets up more per-launch resources, a one-off 20-30 ms host stall -- and the W warm-up steps of a short command never
    # have more than one replay in flight (W = 5 = one 5-step graph), so that stall landed inside the first timed region in
    # about one run out of three (no_preroll 400-1100 instead of 2200 steps/s, gpurun_out/r6f / r6final / r6p; the collector
    # was ruled out: it happens with gc disabled).  Two regions' worth of replays back to back, once, before any timing.
    if not args.eager:
        prime = max(2 * max(args.steps, 1) if args.steps <= 64 else 0, 4 * spg if spg > 1 else 0)  # (>= 4 graphs in flight once)
        if prime:
            wl.run(prime)
            torch.cuda.synchronize()
            n_done += prime
    dt_cold = None
    if args.preroll_ms > 0 and not args.no_cold:
        torch.cuda.synchronize()
        dt_cold = max_over_ranks(timed_steps(wl, args.steps, args.warmup, barrier))
        n_done += args.warmup + args.steps
    pre_ms = preroll(device, args.preroll_ms)
    dt = max_over_ranks(timed_steps(wl, args.steps, args.warmup, barrier))
    n_done += args.warmup + args.steps

    stats = eng.st.read_stats()
    assert all(np.isfinite(v) for v in stats.values()), stats
    assert eng.st.device_step() == n_done

    # ---- the headline measurement is done.  Everything below is side information; under data parallelism it holds
    # collectives (the per-collective probe, the C4 line), i.e. a rank that fails or stalls there would hang its peers and
    # the job would end without its line.  A watchdog guards the line: if the extras are not through after
    # --extras-timeout seconds, rank 0 prints the headline fields alone (marked) and every rank leaves.
    import threading
    core = None
    if rank == 0:
        core = {"metric": "grad-steps/sec", "value": round(world * args.steps / dt, 2), "unit": "grad-steps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": cfg["desc"], "name": args.config, "global_batch": cfg["B"] * world,
                           "parallelism": f"dp{world}", "graph": bool(getattr(eng, "graph", None) is not None)},
                "rccl_ranks": rccl_ranks, "extras": f"timed out after {args.extras_timeout:.0f} s: headline fields only"}

    def bail():
        if core is not None:
            os.write(json_fd, (json.dumps(core) + "\n").encode())
        os._exit(0)

    watchdog = None
    if dp is not None and args.extras_timeout > 0:
        watchdog = threading.Timer(args.extras_timeout, bail)
        watchdog.daemon = True
        watchdog.start()

    # data parallel: how long each of the step's collectives takes inside the step (every rank runs the probe)
    coll = None
    if dp is not None and cfg["algo"] == "cpq":
        try:
            coll = collectives_in_step(eng, dp)
        except Exception as e:
            coll = {"error": repr(e)[:200]}
            torch.cuda.synchronize()

    # N > 1: BASELINE.json's multi-GPU config is C4 (CPQ at (17, 6), 2048 rows per GPU = global batch 16384 at 8 GPUs);
    # every rank runs it (collectives inside), rank 0 reports it under other_configs
    c4_dp = None
    if (world > 1 or force_dp) and args.config == "c2" and not args.no_extras:  # (force_dp: the same code on one rank)
        try:
            w4 = Workload("c4", device, rank, world, dp, n_store=1 << 18, use_graph=not args.eager)
            dt4 = timed_steps(w4.step, 200, 20, barrier)
            import torch.distributed as dist  # noqa: F811
            t4 = torch.tensor([dt4], dtype=torch.float64, device=device)
            dist.all_reduce(t4, op=dist.ReduceOp.MAX)
            dt4 = float(t4.item())
            f4, x4 = flops_per_step(w4.cfg), executed_flops_per_step(w4.cfg, bool(getattr(w4.eng, "ood_rows", False)), _share(w4.eng))
            c4_dp = {"value": round(world * 200 / dt4, 2), "optimizer_steps_per_s": round(200 / dt4, 2),
                     "ms_per_step": round(dt4 / 200 * 1e3, 4), "global_batch": w4.cfg["B"] * world,
                     "parallelism": f"dp{world}", "gflop_per_step_per_gpu": round(f4 / 1e9, 2),
                     "step_frac": round(f4 / (dt4 / 200) / 1e12 / PEAK_FP32_TFLOPS, 4),
                     "step_frac_executed": round(x4 / (dt4 / 200) / 1e12 / PEAK_FP32_TFLOPS, 4),
                     "graph": bool(getattr(w4.eng, "graph", None) is not None)}
            del w4
        except Exception as e:  # the headline line must survive a failing side measurement -- on every rank alike
            c4_dp = {"error": repr(e)[:200]}

    if rank == 0:
        ms = dt / args.steps * 1e3
        fl, fx = flops_per_step(cfg), executed_flops_per_step(cfg, bool(getattr(wl.eng, "ood_rows", False)), _share(wl.eng))
        B = cfg["B"]
        out = {
            "metric": "grad-steps/sec", "value": round(world * args.steps / dt, 2),
            "unit": f"grad-steps/s, one step = one gradient step on one GPU's {B}-row batch; aggregate over the job = "
                    f"n_gpus x optimizer steps/s (each optimizer step consumes a global batch of {B} x n_gpus)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["desc"] + "; on-device minibatch sampling from a HBM-resident synthetic store + "
                                                 "Philox noise inside the step",
                       "name": args.config, "global_batch": B * world, "parallelism": f"dp{world}",
                       "graph": bool(getattr(eng, "graph", None) is not None),
                       # train steps per replayed graph (engine/pipeline.py: step k+1's head under step k's tail); the K
                       # timed steps are K // n such replays + K % n single-step replays
                       "steps_per_graph": spg,
                       # what the shape-keyed plan chooser picked for this workload (engine/plan.py)
                       "plan": (lambda pl: None if pl is None else {k: v for k, v in vars(pl).items()})(getattr(eng, "plan", None)),
                       # both timing protocols where a reader of the driver's record sees them: `value` is
                       # measured behind `preroll_ms` of untimed GEMM work, `no_preroll_value` is the same W + K steps
                       # timed first, straight after the probes
                       "timing": {"preroll_ms": round(pre_ms, 1),
                                  "no_preroll_value": None if dt_cold is None else round(world * args.steps / dt_cold, 2)}},
            "optimizer_steps_per_s": round(args.steps / dt, 2),
            "preroll_ms": round(pre_ms, 1),  # untimed GEMM work queued before the W warm-up steps (see preroll())
            # the same W + K steps timed FIRST, without the pre-roll (rounds 1-3's protocol: the device's power state is
            # still ramping during a 10 ms timed region) -- both protocols in every line
            "no_preroll": None if dt_cold is None else {"value": round(world * args.steps / dt_cold, 2),
                                                       "ms_per_step": round(dt_cold / args.steps * 1e3, 4)},
            "transitions_per_s": round(world * B * args.steps / dt, 1),
            "rccl_ranks": rccl_ranks,
            # what carries the data-parallel step's exchanges: RCCL launches, or IPC-mapped buffers + flags (engine/dist_ipc.py)
            "dp_exchange": None if dp is None else {"kind": exchange, "ranks": world, "one_gpu": one_gpu,
                                                    "status": dp.status() if exchange == "ipc" else None},
            # data parallel only: each collective of the step timed inside the eagerly issued step body (issue order);
            # sum / (1e3 * ms_per_step) = the share of the step spent in exposed communication
            "collectives_in_step": coll,
            # two accountings of the same step: the REFERENCE's work for it (SURVEY.md 8d formula; what a reference
            # user gets per step) and the FLOPs this build actually issues (the reference's discarded VAE decoder on
            # the N*B rows and its repeated actor forwards are not executed) -- hardware utilisation is the second
            "algorithmic_gflop_per_step": round(fl / 1e9, 2),
            "step_tflops": round(fl / (dt / args.steps) / 1e12, 3),
            "step_frac": round(fl / (dt / args.steps) / 1e12 / PEAK_FP32_TFLOPS, 4),
            "step_frac_is": "reference_equivalent (FLOPs of the reference's step / time / fp32 peak)",
            "executed_gflop_per_step": round(fx / 1e9, 2),
            "step_frac_executed": round(fx / (dt / args.steps) / 1e12 / PEAK_FP32_TFLOPS, 4),
            "last_stats": {k: round(float(v), 5) for k, v in stats.items()},
        }
        if roof is not None:
            roof["step_frac"] = out["step_frac"]
            roof["step_frac_executed"] = out["step_frac_executed"]
            out["roofline"] = roof
        out["lease"] = lease
        if c4_dp is not None:
            out["other_configs"] = {"c4": c4_dp}
        if world == 1 and not force_dp and not args.no_extras:
            # the Trainer-API path on an engine of its own, as a reference-style train script gets it (round 6: measured on
            # the engine the pipelined graphs had been built on it runs 8-10 % slower -- 2020-2050 vs 2230-2265 steps/s --
            # with identical kernel timelines under rocprofv3 and an identical host profile: recorded as unexplained in
            # DESIGN_LOG.md round 6; a user of train_one_step() never builds those graphs)
            del wl, eng
            torch.cuda.empty_cache()
            wl_api = Workload(args.config, device, rank, world, None, n_store=1 << 18, use_graph=not args.eager, steps_per_graph=1)
            out["api_path"] = api_path(wl_api)
            del wl_api
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(args.config, device, args.steps_per_graph)
        if world == 1 and not force_dp and not args.no_extras:
            gap = out["cost_return_gap"] = cost_return_gap(device)
            # the metric's second half, compact, where the driver's record keeps it (config is carried over verbatim)
            out["config"]["cost_return_gap_vs_ref"] = {
                k: {"return_gap_max_rel": v.get("return_gap_max_rel"), "cost_gap_max": v.get("cost_gap_max")}
                for k, v in gap.items() if isinstance(v, dict)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if args.config == "c2":
                out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            if "other_configs" in out and not args.no_extras:  # the CPU figure beside each of the other configs' GPU numbers
                for k, v in cpu_baseline_others().items():
                    if k in out["other_configs"] and isinstance(out["other_configs"][k], dict):
                        out["other_configs"][k]["cpu_baseline"] = v
        if watchdog is not None:
            watchdog.cancel()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if watchdog is not None:
        watchdog.cancel()
    if dp is not None:
        import torch.distributed as dist
        barrier()  # nobody tears the communicator down before every rank is through
        dist.destroy_process_group()


def api_path(wl, steps=200, warmup=20):
    """The same step through the boundary a reference train script uses: ``trainer.train_one_step(device tensors)``
    (examples/train/train_cpq.py:143) with lazily materialised statistics -- adds the batch copies into the engine's
    static buffers and the logger bookkeeping to the replayed graph."""
    from osrl_amd.common.logger import DummyLogger
    eng, tr = wl.eng, wl.trainer
    if wl.cfg["algo"] == "cdt":
        eng.attach_store(None)
    else:
        eng.attach_replay(None)
    tr.logger, tr.stats_mode = DummyLogger(), "lazy"
    batch = wl.api_batch()
    dt = timed_steps(lambda: tr.train_one_step(*batch), steps, warmup)
    return {"steps_per_s": round(steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4),
            "what": "trainer.train_one_step(device tensors), stats_mode='lazy', DummyLogger"}


def other_configs(skip: str, device, steps_per_graph: int = 1):
    """Short runs of the other single-GPU BASELINE configs (same step definition), so the driver's line carries them."""
    res = {}
    for name, (steps, warm) in (("c1", (500, 50)), ("c2", (200, 20)), ("c3", (60, 10)), ("c4", (200, 20)), ("c5", (10, 3))):
        if name == skip:
            continue
        try:
            w = Workload(name, device, 0, 1, None, n_store=1 << 18, steps_per_graph=steps_per_graph)
            if w.pipe is not None:
                # graph priming, as in front of the headline regions: the first time several replays of a graph are in
                # flight at once the runtime stalls once (20-60 ms) -- and a warm-up shorter than a few graphs never has more
                # than one (C4 at 8 steps per graph: 1435 instead of 2418 steps/s in one run of two, gpurun_out/r6bfinal)
                w.run(4 * w.pipe.n)
                torch.cuda.synchronize()
            dt = timed_steps(w, steps, warm)
            fl = flops_per_step(w.cfg)
            fx = executed_flops_per_step(w.cfg, bool(getattr(w.eng, "ood_rows", False)), _share(w.eng))
            res[name] = {"steps_per_s": round(steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4),
                         "gflop_per_step": round(fl / 1e9, 2),
                         "step_frac": round(fl / (dt / steps) / 1e12 / PEAK_FP32_TFLOPS, 4),
                         "executed_gflop_per_step": round(fx / 1e9, 2),
                         "step_frac_executed": round(fx / (dt / steps) / 1e12 / PEAK_FP32_TFLOPS, 4),
                         "steps_per_graph": w.pipe.n if w.pipe is not None else 1}
            del w
            torch.cuda.empty_cache()
        except Exception as e:  # a failing side measurement must not take the headline line down
            res[name] = {"error": repr(e)[:200]}
    return res


if __name__ == "__main__":
    main()
