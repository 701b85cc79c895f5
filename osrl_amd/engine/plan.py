"""Launch-plan choices in ONE place: a shape-keyed chooser with the BASELINE configs' rows pinned, and
the registry of every tuning switch the engines know.

Two kinds of thing used to be spread over ~35 ``os.environ`` reads in engine/*.py:

  * **shape rules** -- what a step plan does differently for (obs_dim, act_dim, batch, widths): which dW tile a group
    takes, whether the action draws ride on the actor trunks' forward launch, whether the VAE phase runs as all-CU layer
    launches, ...  They are functions of the shape, calibrated on the BASELINE configs (reference configs:
    examples/configs/cpq_configs.py:27-50, bcql_configs.py:31-51).  ``cpq_plan`` / ``bcql_plan`` compute them,
    ``PINNED`` states what they must return for C2 / C3 / C4 (tests/test_host_cpu.py holds the chooser to it), and
    tests/test_gpu_train_step.py::test_random_shape_tuples_match_the_oracle sweeps random tuples through whatever plan
    the chooser picks, against the oracle -- so no reachable plan is untested.
  * **lab switches** -- A/B knobs of the kernel work (``OSRL_*``).  ``knob()`` registers each with its default and a
    one-line meaning and reads the environment ONLY when ``OSRL_LAB=1``: a production run's plan does not depend on
    stray variables, a lab run (tools/gpu_*.sh export OSRL_LAB=1) can still flip them.  ``python -m osrl_amd.engine.plan``
    prints the registry and the pinned rows.

Operator switches that are not plan choices stay plain environment variables: OSRL_LIB (alternative library build),
OSRL_DP_EAGER (run the data-parallel step without capturing its collectives), OSRL_FORCE_DP (bench.py), OSRL_LAB -- and
the four safety switches registered with ``knob(..., operator=True)``: OSRL_ARG_ARENA, OSRL_BC_ONE_LAUNCH,
OSRL_FUSE_DW_ADAM, OSRL_SEEDS (bit-equal fallback forms, honoured without OSRL_LAB).
"""
from __future__ import annotations

import os
from dataclasses import asdict, dataclass
from typing import Dict, Optional, Tuple

KNOBS: Dict[str, Tuple[str, str]] = {}
_warned = set()


def lab() -> bool:
    return os.environ.get("OSRL_LAB") == "1"


def knob(name: str, default: str, doc: str = "", operator: bool = False) -> str:
    """The value of switch ``name``: its default, or what the environment says -- for a LAB switch only under OSRL_LAB=1;
    ``operator=True`` marks a safety switch (a bit-equal alternative form an operator may need to fall back to:
    OSRL_ARG_ARENA, OSRL_BC_ONE_LAUNCH, OSRL_FUSE_DW_ADAM, OSRL_SEEDS), which is honoured in every run like OSRL_DP_EAGER
    (ADVICE r5).  An ignored lab switch is reported once per process on stderr (``warnings.warn`` is deduplicated or
    filtered by many launchers)."""
    KNOBS.setdefault(name, (default, doc + (" [operator switch]" if operator else "")))
    if name in os.environ:
        if operator or lab():
            return os.environ[name]
        if name not in _warned and os.environ[name] != default:
            _warned.add(name)
            import sys
            print(f"osrl_amd: {name}={os.environ[name]} is IGNORED -- lab switches are read only under OSRL_LAB=1 "
                  f"(using {default!r}); operator switches: OSRL_ARG_ARENA, OSRL_BC_ONE_LAUNCH, OSRL_FUSE_DW_ADAM, OSRL_SEEDS, "
                  f"OSRL_DP_EAGER", file=sys.stderr, flush=True)
    return default


def knob_set(name: str) -> bool:
    """True when a lab run set ``name`` explicitly (switches whose default is "derive it from the shape")."""
    return lab() and name in os.environ


@dataclass(frozen=True)
class CPQPlan:
    """What the CPQ step plan derives from the shape (engine/cpq.py reads these, nothing else)."""
    head_tails: bool          # every action draw of the step by the actor trunks' forward launch (N * act_dim <= 32)
    vae_dw_tile: int          # dW tile of the VAE group in 16-blocks: 5 (80 x 80, 400-wide layers are 5 x 5) or 0 (default)
    vae_dw_splits: int        # row splits of that plan
    small_dw: bool            # critic / cost-critic dW on 32 x 32 tiles x 2 splits (fit beside the N*B encoder launch)
    ood_tile: int             # row tile of the N*B-row launches (80 = one workgroup per CU)
    vae_ns: bool              # the VAE phase as all-CU layer launches (csrc/vae_ns.hip)
    vae_adam_side: bool       # single GPU: the VAE's optimizer step at the head of the side branch's second half
    steps_per_graph: int      # engine.steps_replay(): train steps per replayed hipGraph (engine/pipeline.py); 1 = one step
    ood_rows: bool = False    # single GPU: the target cost critics of the OOD penalty on the SELECTED rows only (cpq.py:183-184)
    ood_rows_late: bool = True   # ... and the side branch's second half then starts behind the cost critics' dW launch
    ood_share: bool = False   # the two N*B-row launches on tiles of shared observations: the observation part of layer 0 once
    #                           per observation of a tile (osrl_rows_t.share0)
    pipe_no_join: bool = False   # pipelined graphs: no join between the steps of a graph (the dual step of step k at the head
    #                              of step k+1's side branch)
    pipe_prologue: str = "early"  # pipelined graphs: where step k+1's prologue sits on step k's side branch: in front of the
    #                              OOD statistic ("early") / in front of the critic phase ("critic") or first on the branch
    #                              ("head"): covered by the main chain's wait for the critic's Adam


def ood_rows_ok(od: int, ad: int, B: int, N: int, c_hidden) -> bool:
    """Shapes on which the row-set form of the N*B-row target-cost-critic launch exists (csrc/mlp_nb.hip, LIST
    instantiation of the 4-wave 80-row kernel): every hidden layer 13..16 column blocks wide, input <= 128 columns, the
    KL keys in one workgroup's registers."""
    return (c_hidden is not None and len(c_hidden) >= 1 and all(13 <= (int(h) + 15) // 16 <= 16 for h in c_hidden)
            and od + ad <= 128 and N * B <= 32768)


def cpq_plan(od: int, ad: int, B: int, vae_hidden: int, N: int, seeds: bool = True, c_hidden=(256, 256)) -> CPQPlan:
    head_tails = {"1": True, "0": False}.get(knob("OSRL_HEAD_TAILS", "auto", "action draws as forward tails: 1 / 0 / auto"),
                                             N * ad <= 32)
    t5 = knob("OSRL_VAE_DW_T5", "1", "VAE dW on 80 x 80 tiles where the width allows") == "1" and vae_hidden % 80 == 0 and B >= 1024
    splits = int(knob("OSRL_VAE_DW_SPLITS", "0", "row splits of the VAE dW plan (0 = by rule)")) or \
        (max(1, (3 * B) // 2048) if t5 else max(1, B // 1024))
    ns_mode = knob("OSRL_VAE_NS", "auto", "VAE phase as all-CU layer launches: 1 / 0 / auto")
    ns_shape = vae_hidden % 80 == 0 and 80 <= vae_hidden <= 448 and ad <= 8 and od + 2 * ad <= 128
    vae_ns = seeds and ns_shape and (ns_mode == "1" or (ns_mode == "auto" and vae_ns_auto(B, od, ad, vae_hidden)))
    # the VAE's Adam off the main chain (its only reader this step is the side branch's N*B-row encoder launch): C2 +0.7 %
    # with the fused VAE launches (gpurun_out/r5h), +1.1 % with the all-CU ones (r5k2).  At C4 the side branch is the longer
    # one -- its action draws are four launches of their own there (head_tails off) -- and the same move costs 3 % (2340 vs
    # 2416, gpurun_out/r5d): the optimizer step goes where the draws ride on the actor launch
    side = {"1": True, "0": False}.get(knob("OSRL_VAE_ADAM_SIDE", "auto", "VAE Adam on the side branch: 1 / 0 / auto"),
                                       B >= 1024 and bool(head_tails))
    vt = int(knob("OSRL_VAE_DW_TILE", "0", "dW tile of the VAE group in 16-blocks (0 = by rule)")) or (5 if t5 else 0)
    # several steps per graph with the next step's prologue on the side branch (engine/pipeline.py, round 6): C2 +1.2 .. +2.6 %
    # at 4 steps per graph over 200 steps (2335-2365 vs 2305, gpurun_out/r6e); on the driver's K = 20 / W = 5 command, medians of
    # five alternating rounds: 2293 (1) / 2308 (2) / 2317 (4) / 2329 (5) / 2328 (10) (profiles/r6_k20_steps_per_graph.txt) -> 5.
    # C4 within +-1 % of one step per graph -- its side branch is the longer one (separate action-draw launches) and gets the
    # extra prologue: on where the draws ride on the actor launch
    # Second session of round 6: graphs whose steps are NOT joined -- step k's dual step at the head of step k+1's side branch
    # (it has no reader but itself), the next prologue early enough on the side branch that the main chain's wait for the
    # critic's Adam covers it, so the next step's first VAE launch follows the actor group's Adam with no packet of its own
    # in between (un-profiled start stamps, tools/trace_steps.py: 1.3 instead of 10 us between that Adam's end and the launch).
    # WHERE the 14 us prologue sits decides how the N*B-row launch behind it lines up with the main chain's all-CU VAE launches,
    # and that is worth more than the boundary itself (DESIGN_LOG): un-profiled clocks, alternating rounds --
    #   C2 (draws ride on the actor launch; gpurun_out/r6nj4, r6nj5: 16 + 24 runs): prologue FIRST on the side branch ("head")
    #      2356-2379 at K = 300 / 2342-2376 on the driver's K = 20 command against 2324-2328 / 2293-2345 joined: +1.8 %; in front
    #      of the critic phase 2263-2283, in front of the OOD statistic with an event of its own 2230 (both LOSE: the VAE phase
    #      stretches from 163 to 190 us when the N*B-row launch starts 6 us earlier against it);
    #   C4 (separate draw launches, the side branch is the longer chain; gpurun_out/r6nj3, r6trace): in front of the critic
    #      phase 2411-2417 at 4 steps per graph against 2330-2360 joined, 2361-2368 at one step per graph, 2358-2366 with "head": +2 %.
    # Steps per graph under the no-join plans (a replay boundary is the one full join left; gpurun_out/r6nj6, r6nj7): C2 at
    # K = 300 2358-2373 (5) / 2355-2378 (10) / 2390-2400 (20); on the driver's K = 20 command, six alternating runs, medians 2362
    # (5) / 2383 (10) / 2390 (20: the 20 timed steps are ONE replay) -> 20.  C4 2399-2404 (2) / 2408-2436 (4) / 2423-2443 (8) -> 8; under
    # the third session's plan (ood_rows, the ordering edges; gpurun_out/r6spg2, three alternating rounds at K = 320): 2480-2488 (8) /
    # 2497-2522 (12) / 2505-2540 (16) / 2526-2535 (20) -> 20 there too.
    side_long = not head_tails and B >= 1024
    spg = int(knob("OSRL_PIPE_STEPS", "0", "train steps per pipelined graph (0 = by rule)")) or \
        (20 if B >= 1024 else 1)
    dual = knob("OSRL_PIPE_DUAL", "auto", "pipelined steps: the dual step behind the join (main) / on the side branch behind the "
                "OOD statistic (side) / at the head of the NEXT step's side branch, no join between the steps of a graph (next)")
    pro = knob("OSRL_PIPE_PROLOGUE", "auto", "pipelined steps: the next step's prologue behind the OOD statistic (side) / in "
               "front of it (early) / on the main chain (main) / on the side branch in front of the critic phase (critic) or "
               "first on it (head): the main chain's wait for the critic's Adam then covers it")
    no_join = B >= 1024 if dual == "auto" else dual == "next"
    pro = ("critic" if side_long else "head" if B >= 1024 else "early") if pro == "auto" else pro
    ood_tile = int(knob("OSRL_OOD_TILE", "80", "row tile of the N*B-row launches (0 = 32-row tile loop)"))
    # qc_ood = ((KL >= quantile(KL, 0.75)) * qc_sampled).mean(0) (cpq.py:183-184) multiplies three quarters of the N*B target
    # cost-critic outputs by zero: with the encoder launch, the quantile and a compaction in FRONT of that forward it runs on
    # the selected quarter only -- 5.3 of C2's 29.6 issued GFLOP per step gone.  Built, parity-tested (same network
    # parameters bit for bit) and measured at the end of round 6: C2 +0.3 %, C4 +1.3 % (profiles/r6_ood_rows_ab.txt) -- 18 % of
    # the FLOPs buy one per cent because the step is not FLOP-bound; the forward moves from the idle
    # early part of the side branch to its tail, behind a 30-50 us single-workgroup select (that first form forced the joined graph).
    # Third session of round 6, inside the no-join graphs (the cost critics' target update of step k carried to the head of
    # step k+1's side branch, behind its last reader), three changes together make it the rule: (1) it no longer forces the
    # joined graph; (2) ``ood_rows_late``: the second half of the side branch waits for the cost critics' dW launch instead of
    # the VAE's -- without the N*B-row cost-critic launch in front of it the branch reaches the N*B-row encoder launch ~70 us
    # earlier, beside that dW launch, whose 384 small workgroups then take 70 instead of 25 us; (3) the encoder launch keeps
    # its shared-observation tiles (only the row SET has none).  Alternating rounds, un-profiled clocks:
    #   C4 2498-2513 (1) / 2528-2541 (1 + 2) / 2527-2552 (1 + 2 + 3) against 2432-2443: +4.3 % (gpurun_out/r6oodrows2, 4, 5);
    #   C2 2400-2433 (1) / 2441-2469 (1 + 2) / 2458-2475 (all three, five rounds at K = 300) against 2395-2420: +2.3 %; on the
    #      driver's K = 20 command 2462-2499 (median 2490) against 2404-2459 (median 2433) (gpurun_out/r6oodrows6).
    # Rule: on wherever the graphs are not joined and the row-set kernel exists.
    rows_mode = knob("OSRL_OOD_ROWS", "auto", "target cost critics on the selected OOD rows only (single GPU): 1 / 0 / auto")
    ood_rows = (rows_mode == "1" or (rows_mode == "auto" and bool(no_join))) \
        and ood_tile == 80 and ood_rows_ok(od, ad, B, N, c_hidden)
    rows_late = knob("OSRL_OOD_ROWS_LATE", "1", "plan.ood_rows: the side branch's second half waits for the cost critics' dW "
                     "launch instead of the VAE's: 1 / 0") == "1"
    # the N*B rows of the two OOD launches are the B observations N times over (cpq.py:164-176): on tiles of [5 copies] x [16
    # observations] the observation columns of layer 0 are multiplied once per observation (osrl_rows_t.share0, csrc/mlp_nb.hip
    # nb_share_acc; second session of round 6).  Isolated launches at C2: cost critics 69.4 -> 60.9 us, encoder 72.1 -> 65.7;
    # step, three alternating rounds: C2 2430-2448 against 2386-2395 (+2.3 %), C4 (one of two k-steps) 2419-2445 against 2411-2448
    # (gpurun_out/r6share).  The outputs differ from the plain launch by fp32 rounding (another order of a row's sum); they
    # feed qc_ood / the KL quantile only -- no gradient to any network (cpq.py:155-186) -- so parameters are the same bits.
    ood_share = knob("OSRL_OOD_SHARE", "1", "N*B-row launches on tiles of shared observations (observation part of layer 0 once per observation): 1 / 0") == "1" \
        and ood_tile == 80 and od >= 16 and B % 16 == 0 and N % 5 == 0
    return CPQPlan(head_tails=bool(head_tails), vae_dw_tile=vt, vae_dw_splits=splits, small_dw=B >= 1024,
                   ood_tile=ood_tile, vae_ns=bool(vae_ns), vae_adam_side=bool(side), steps_per_graph=spg,
                   ood_rows=bool(ood_rows), ood_rows_late=bool(rows_late), ood_share=bool(ood_share), pipe_no_join=bool(no_join), pipe_prologue=pro)


def vae_ns_auto(rows: int, od: int, ad: int, vae_hidden: int = 400) -> bool:
    """Where the five all-CU VAE launches beat the four fused ones INSIDE the step (A/B on MI355X, DESIGN_LOG round 5):
    C4's (17, 6) at 2048 rows +4.5 % (gpurun_out/r5a); C2's (76, 2) at 2048 rows +0.2 % on their own but +1.1 % together
    with the VAE's Adam on the side branch (three pairs, gpurun_out/r5k2: 2280 / 2274 / 2282 vs 2250 / 2257 / 2252); C3's
    (33, 8) at 4096 rows -3.6 % (two rounds of 48-row tiles against one round of 256 fused 16-row tiles).  The rule is
    those points, not a model: one round of tiles, at the ONE hidden width that was timed (400 = 5 column groups; other
    widths are parity-tested -- tests/test_gpu_kernels.py test_vae_ns_launches_equal_the_fused_launches covers 80 / 160 /
    240 / 320 / 400 -- but nobody timed them, so they stay on the fused launches unless OSRL_VAE_NS=1 asks).
    A shape outside the calibrated points can be timed at engine construction: OSRL_LAB=1 OSRL_PLAN_TIME=1 runs both VAE
    forms once and logs which won (engine/cpq.py time_vae_forms)."""
    return 1024 <= rows <= 2048 and vae_hidden == 400


@dataclass(frozen=True)
class BCQLPlan:
    vae_dw_tile: int
    target_tile: int          # row tile of the N*B-row target pipelines (80 = mlp_fwd_nb_kernel)
    vae_ns: bool
    dw_splits: int            # row splits of the critic / cost-critic / actor dW plans (0 = DwPlan's rule)
    steps_per_graph: int      # engine.steps_replay(): train steps per replayed hipGraph (engine/pipeline.py); 1 = one step


def bcql_plan(od: int, ad: int, B: int, vae_hidden: int, N: int, seeds: bool = True) -> BCQLPlan:
    t5 = knob("OSRL_VAE_DW_T5", "1") == "1" and vae_hidden % 80 == 0 and B >= 1024
    ns_mode = knob("OSRL_VAE_NS", "auto")
    ns_shape = vae_hidden % 80 == 0 and 80 <= vae_hidden <= 448 and ad <= 8 and od + 2 * ad <= 128
    # (BCQ-Lag: the all-CU VAE launches only on request -- the one measurement there is C3's 4096 rows, -3.6 %)
    vae_ns = seeds and ns_shape and ns_mode == "1"
    # DwPlan's rule aims at ~2048 (tile, split) workgroups: 16 splits of 256 rows for the 48 tiles of a twin ensemble at
    # 4096 rows = 768 workgroups, one and a half rounds of the 512 resident slots, 16 k-steps each.  6 splits (288
    # workgroups of 43 k-steps, one round): C3 648-650 vs 639-642 steps/s, the same at 4 and 8 (gpurun_out/r5n2)
    dws = int(knob("OSRL_BCQ_DW_SPLITS", "0", "BCQ-Lag: row splits of the critic / cost-critic / actor dW plans (0 = by rule)")) \
        or (6 if B >= 4096 else 0)
    # several steps per graph, the next step's prologue + VAE phase on the side queue under this step's actor phase
    # (engine/pipeline.py, round 6): C3 651 -> 659-663 at 4 steps per graph, 670 at 10 (gpurun_out/r6b) -- the overlapped
    # phases are throughput-bound (450 us together against 205 + 222 alone, profiles/r6_timeline_4step_c3.txt), what is
    # gained are the boundary's bubbles; measured at 4096 rows only
    spg = int(knob("OSRL_PIPE_STEPS", "0")) or (10 if B >= 4096 else 1)
    return BCQLPlan(vae_dw_tile=5 if t5 else 0,
                    target_tile=int(knob("OSRL_BCQ_TILE", "80", "row tile of BCQ-Lag's N*B-row target pipelines")),
                    vae_ns=bool(vae_ns), dw_splits=dws, steps_per_graph=spg)


# BASELINE.json configs -> the plan the chooser must give (tests/test_host_cpu.py::test_plan_rows_are_pinned); a changed
# rule that moves one of these rows is a deliberate act with a measurement behind it (DESIGN_LOG)
PINNED = {
    "c2": (cpq_plan, dict(od=76, ad=2, B=2048, vae_hidden=400, N=10),
           CPQPlan(head_tails=True, vae_dw_tile=5, vae_dw_splits=3, small_dw=True, ood_tile=80, vae_ns=True, vae_adam_side=True,
                   steps_per_graph=20, ood_rows=True, ood_share=True, pipe_no_join=True, pipe_prologue="head")),
    "c4": (cpq_plan, dict(od=17, ad=6, B=2048, vae_hidden=400, N=10),
           CPQPlan(head_tails=False, vae_dw_tile=5, vae_dw_splits=3, small_dw=True, ood_tile=80, vae_ns=True, vae_adam_side=False,
                   steps_per_graph=20, ood_rows=True, ood_share=True, pipe_no_join=True, pipe_prologue="critic")),
    "c3": (bcql_plan, dict(od=33, ad=8, B=4096, vae_hidden=400, N=10),
           BCQLPlan(vae_dw_tile=5, target_tile=80, vae_ns=False, dw_splits=6, steps_per_graph=10)),
    "cpq_small": (cpq_plan, dict(od=5, ad=2, B=16, vae_hidden=48, N=4, c_hidden=(32, 32)),
                  CPQPlan(head_tails=True, vae_dw_tile=0, vae_dw_splits=1, small_dw=False, ood_tile=80, vae_ns=False, vae_adam_side=False,
                           steps_per_graph=1, ood_rows=False)),
}


def describe(plan) -> Dict[str, object]:
    return asdict(plan)


if __name__ == "__main__":  # pragma: no cover
    import importlib
    for mod in ("core", "glue", "cpq", "bcql", "bc", "cdt"):  # importing the engines registers their knobs
        try:
            importlib.import_module(f"osrl_amd.engine.{mod}")
        except Exception as e:  # no library on this host: the registry of the modules that did import is still printed
            print(f"({mod}: {e!r})")
    for name, (fn, kw, want) in PINNED.items():
        print(f"{name:10s} {kw} -> {fn(**kw)}")
    for k, (d, doc) in sorted(KNOBS.items()):
        print(f"{k:24s} default {d!r:8s} {doc}")
