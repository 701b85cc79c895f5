"""Knock-out pricing of the captured CPQ step (lab, timing only): which launch is worth how much of the step.

For each launch (or launch group) of ``CPQEngine.body`` the bench workload is re-captured WITHOUT it and timed: the step's
numbers are meaningless then, its duration is not -- `step - step_without(X)` is what X costs the step where it stands
(its own time minus whatever it overlaps), the upper bound of anything done to X.  Written after round 5's schedule work:
the step is two full graph branches, so a launch's duration in the timeline says little about its price.

    OSRL_LAB=1 python tools/r5_knockout.py [c2|c4] > gpurun_out/knockout.txt
"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from osrl_amd.engine import glue as G  # noqa: E402


@contextlib.contextmanager
def patched(pairs):
    saved = []
    try:
        for obj, name, fn in pairs:
            saved.append((obj, name, obj.__dict__.get(name, None), name in obj.__dict__))
            setattr(obj, name, fn)
        yield
    finally:
        for obj, name, old, had in reversed(saved):
            if had:
                setattr(obj, name, old)
            else:
                delattr(obj, name)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    dev = torch.device("cuda", 0)
    wl = bench.Workload(cfg, dev, 0, 1, None, n_store=1 << 18)
    e = wl.eng

    def timed():
        e.graph = None
        ts = [bench.timed_steps(wl.step, 200, 20) / 200 * 1e6 for _ in range(3)]
        return float(np.median(ts))

    nothing = lambda *a, **k: None  # noqa: E731
    upd, opt = e._update, e._optim

    def update_without(group):
        return lambda name, tau: None if name == group else upd(name, tau)

    def optim_without(group, what):
        def f(name, plan, tau):
            if name != group:
                return opt(name, plan, tau)
            if what == "adam":  # dW only
                plan.launch()
            elif what == "dw":  # Adam only
                upd(name, tau)
        return f

    cases = {
        "VAE forward (3 all-CU launches)": [(e.vae_ns, "forward", nothing)] if e.vae_ns is not None else None,
        "VAE backward (2 all-CU launches)": [(e.vae_ns, "backward", nothing)] if e.vae_ns is not None else None,
        "VAE dW": [(e.p_vae, "launch", nothing)],
        "VAE Adam": [(e, "_update", update_without("vae")), (e, "_optim", optim_without("vae", "adam"))],
        "actor forwards on obs' and obs (+ action draws)": [(e.r_actor_next, "forward_with",
                                                            lambda *a, **k: (e.r_actor_next.y, e.r_actor_obs.y))],
        "N*B-row target cost critics": [(e.r_costold_ood, "forward", lambda *a, **k: e.r_costold_ood.y)],
        "critic phase forward (targets + online)": [(e.r_old_next, "forward_with", lambda *a, **k: (e.r_old_next.y, e.r_critic.y))],
        "critic backward dZ": [(e.r_critic, "backward_dz", nothing)],
        "critic dW": [(e, "_optim", optim_without("critic", "dw"))],
        "critic Adam": [(e, "_optim", optim_without("critic", "adam"))],
        "cost-critic phase forward": [(e.r_costold_next, "forward_with", lambda *a, **k: (e.r_costold_next.y, e.r_cost.y))],
        "cost-critic backward dZ": [(e.r_cost, "backward_dz", nothing)],
        "cost-critic dW + Adam": [(e.p_cost, "launch_adam", nothing), (e.p_cost, "launch", nothing),
                                  (e, "_update", update_without("cost_critic"))],
        "N*B-row VAE encoder (+ KL rows)": [(e.r_enc_ood, "forward", lambda *a, **k: e.r_enc_ood.y)],
        "quantile + OOD mean": [(G, "cpq_ood_stat", nothing)],
        "actor-phase Q forward": [(e.r_pi_q, "forward", lambda *a, **k: e.r_pi_q.y)],
        "actor-phase Q backward": [(e.r_pi_q, "backward_dz", nothing)],
        "actor backward dZ": [(e.r_actor_obs, "backward_dz", nothing)],
        "actor dW": [(e, "_optim", optim_without("actor", "dw"))],
        "actor Adam": [(e, "_optim", optim_without("actor", "adam"))],
        "dual step": [(G, "cpq_alpha_step", nothing)],
    }
    groups = {
        "EVERYTHING (the prologue + the graph's edges remain)": list(cases),
        "everything but the two N*B-row launches": [k for k in cases if not k.startswith("N*B")],
        "the two N*B-row launches": [k for k in cases if k.startswith("N*B")],
        "the VAE phase (forward, backward, dW, Adam)": [k for k in cases if k.startswith("VAE")],
        "the actor phase (Q forward / backward, actor backward, dW, Adam)": [k for k in cases if k.startswith("actor-phase") or k in ("actor backward dZ", "actor dW", "actor Adam")],
    }
    base = [timed(), timed()]
    print(f"{cfg}: step {base[0]:.1f} / {base[1]:.1f} us ({1e6 / np.mean(base):.0f} steps/s), plan {e.plan}")
    b = float(np.mean(base))
    rows = []
    for name, pairs in cases.items():
        if pairs is None:
            continue
        try:
            with patched(pairs):
                t = timed()
            rows.append((b - t, name, t))
        except Exception as ex:  # a knock-out the engine refuses (a later launch validates its inputs)
            print(f"  {name}: {ex!r}"[:160])
    for gname, keys in groups.items():
        pairs, seen = [], set()
        for k in keys:
            for pr in (cases[k] or []):
                if (id(pr[0]), pr[1]) in seen:  # (two cases patch the same method: _optim / _update -- take no-ops)
                    continue
                seen.add((id(pr[0]), pr[1]))
                pairs.append(pr)
        if any(pr[1] in ("_optim", "_update") for pr in pairs):
            groups_optim = {"VAE": "vae", "actor": "actor", "critic": "critic", "cost": "cost_critic"}
            drop = {v for k_, v in groups_optim.items() if any(kk.startswith(k_) for kk in keys)}
            pairs = [pr for pr in pairs if pr[1] not in ("_optim", "_update")]
            pairs.append((e, "_optim", lambda name, plan, tau: None if name in drop else opt(name, plan, tau)))
            pairs.append((e, "_update", lambda name, tau: None if name in drop else upd(name, tau)))
        try:
            with patched(pairs):
                t = timed()
            print(f"  {b - t:7.1f} us  without {gname}  ({t:.1f})")
        except Exception as ex:
            print(f"  {gname}: {ex!r}"[:160])
    e.graph = None
    for d, name, t in sorted(rows, reverse=True):
        print(f"  {d:7.1f} us  without {name}  ({t:.1f})")
    print(f"  (sum of the prices {sum(r[0] for r in rows):.0f} us of a {b:.0f} us step; repeat of the unpatched step: {timed():.1f})")


if __name__ == "__main__":
    main()
