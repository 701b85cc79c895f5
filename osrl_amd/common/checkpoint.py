"""Checkpoint round trip.

The reference saves ``{"model_state": model.state_dict()}`` through its logger's ``checkpoint_fn``
(examples/train/train_cpq.py:106-109) and reloads it with ``torch.load`` + ``model.load_state_dict(ckpt["model_state"])``
(osrl/common/exp_util.py:51-74, examples/eval/eval_cpq.py).  ``model_state`` here has the same keys and shapes, so
reference checkpoints load into the build and the build's checkpoints load into the reference.

What the reference cannot do is RESUME training: optimizer moments, the step count, ``log_alpha`` (a plain tensor,
cpq.py:93), the PID controller's integrator (python attributes, net.py:373-374) and CDT's ``log_temperature``
(cdt.py:144-145) are not in ``state_dict``.  The build stores them under a second key, ``"osrl_amd"``, which the
reference's loader ignores:

    {"model_state": {...reference layout...},
     "osrl_amd": {"version": 1, "algo": "CPQ", "step": 1234,
                  "optim": {group: {"exp_avg": {param_key: tensor}, "exp_avg_sq": {...}}},
                  "scalars": {"log_alpha" | "pid_state" | "log_temperature" | "temperature_moments" |
                              "scalar_leaves" (COptiDICE tau / lmbda + moments): tensor}}}

The device noise streams are functions of (seed, step), so a resumed run draws the noise the uninterrupted run
would have drawn: save -> load -> continue is bit-identical to not stopping (tests/test_gpu_train_step.py).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

VERSION = 1
_SCALARS = ("log_alpha", "pid_state", "log_temperature", "scalar_leaves")


def engine_handoff(model, new_engine, old_engine) -> None:
    """The train-step count (Adam bias correction, LR warm-up, Philox offsets) and CDT's temperature moments
    belong to the MODEL's training state, not to one batch size's launch plan: hand them to a rebuilt engine."""
    # the engine's dW plans are complete here: record the slab epochs they were built against NOW, not at the engine's
    # first step -- an engine built, not stepped, and then superseded must be flagged stale
    from ..engine.core import slab_epochs
    new_engine._slab_epochs = slab_epochs(model)
    if old_engine is not None:
        step = old_engine.st.device_step()
    else:
        step = int(getattr(model, "_resume_step", 0))
    new_engine.st.set_step(step)
    if hasattr(new_engine, "temp_mv"):
        src = getattr(old_engine, "temp_mv", None) if old_engine is not None else getattr(model, "_resume_temp_mv", None)
        if src is not None:
            new_engine.temp_mv.copy_(torch.as_tensor(src, dtype=torch.float32))
    dp = getattr(new_engine, "dist", None)
    if dp is not None:  # data parallel: every replica starts from rank 0's state (weights, moments, scalars, step count)
        dp.broadcast_model(model, new_engine)


def train_step_count(model) -> int:
    eng = getattr(model, "_engine", None)
    return eng.st.device_step() if eng is not None else int(getattr(model, "_resume_step", 0))


def checkpoint_state(model, with_optimizer: bool = True) -> Dict[str, Any]:
    """The dict to hand to ``torch.save`` (host tensors only)."""
    chk = getattr(getattr(model, "_engine", None), "check_health", None)
    if chk is not None:
        chk()  # never save parameters a failed one-launch step left behind (engine/bc.py)
    out: Dict[str, Any] = {"model_state": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}
    if not with_optimizer:
        return out
    scalars = {k: getattr(model, k).detach().cpu().clone() for k in _SCALARS
               if isinstance(getattr(model, k, None), torch.Tensor)}
    eng = getattr(model, "_engine", None)
    mv = getattr(eng, "temp_mv", None) if eng is not None else getattr(model, "_resume_temp_mv", None)
    if mv is not None:
        scalars["temperature_moments"] = torch.as_tensor(mv).detach().cpu().clone()
    out["osrl_amd"] = dict(version=VERSION, algo=type(model).__name__, step=train_step_count(model),
                           optim={name: g.optim_state() for name, g in model.groups.items()}, scalars=scalars)
    return out


def save_checkpoint(model, path: str, with_optimizer: bool = True) -> None:
    torch.save(checkpoint_state(model, with_optimizer), path)


def load_checkpoint(model, ckpt, resume: bool = True, strict: bool = True) -> Dict[str, Any]:
    """``ckpt``: a path or an already loaded dict.  Loads ``model_state`` (reference or build checkpoints alike);
    with ``resume`` and an ``"osrl_amd"`` section also the optimizer moments, the step count and the scalar state.
    Returns the loaded dict."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
    if "model_state" not in ckpt:
        raise KeyError('checkpoint has no "model_state" entry (examples/train/train_*.py save {"model_state": ...})')
    model.load_state_dict(ckpt["model_state"], strict=strict)
    extra: Optional[dict] = ckpt.get("osrl_amd")
    if not resume or extra is None:
        return ckpt
    if extra.get("version") != VERSION or extra.get("algo") != type(model).__name__:
        raise ValueError(f"resume section is {extra.get('algo')} v{extra.get('version')}, "
                         f"the model is {type(model).__name__} v{VERSION}")
    for name, g in model.groups.items():
        g.load_optim_state(extra["optim"][name])
    for k, v in extra["scalars"].items():
        if k == "temperature_moments":
            model._resume_temp_mv = v.clone()
        else:
            getattr(model, k).copy_(v)
    model._resume_step = int(extra["step"])
    eng = getattr(model, "_engine", None)
    if eng is not None:  # a live launch plan keeps its buffers; only its counters move
        eng.st.set_step(model._resume_step)
        if hasattr(eng, "temp_mv") and "temperature_moments" in extra["scalars"]:
            eng.temp_mv.copy_(extra["scalars"]["temperature_moments"])
    return ckpt
