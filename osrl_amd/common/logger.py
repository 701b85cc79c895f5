"""Logger stand-in + lazily materialised statistics.

The reference logs through fsrl's ``WandbLogger`` / ``DummyLogger`` (un-vendored; call sites
cpq.py:310-313, train_cpq.py:161-164 define the API: ``store(tab=None, **scalars)``,
``write(step, display)``, ``write_without_reset(step)``, ``save_config``, ``setup_checkpoint_fn``,
``save_checkpoint``).  Any object with ``store`` works as ``logger``; this ``DummyLogger`` keeps a
running mean per key so examples and tests have something to read.
"""
from __future__ import annotations

from typing import Dict, List


class LazyStat:
    """A float-like handle on one logged statistic of one train step.  It reads the device-side
    statistics ring (one sync) only when the value is actually needed, so ``train_one_step`` never
    blocks on ``.item()`` the way the reference does 5-6 times per step."""

    __slots__ = ("_st", "_step", "_key", "_val")

    def __init__(self, st, step: int, key: str):
        self._st, self._step, self._key, self._val = st, step, key, None

    def materialize(self) -> float:
        if self._val is None:
            self._val = float(self._st.read_stats(self._step)[self._key])
            self._st = None
        return self._val

    def __float__(self):
        return self.materialize()

    def __repr__(self):
        return f"LazyStat({self._key}@{self._step})" if self._val is None else repr(self._val)

    def __add__(self, o):
        return float(self) + float(o)

    __radd__ = __add__

    def __sub__(self, o):
        return float(self) - float(o)

    def __rsub__(self, o):
        return float(o) - float(self)

    def __mul__(self, o):
        return float(self) * float(o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return float(self) / float(o)

    def __lt__(self, o):
        return float(self) < float(o)

    def __gt__(self, o):
        return float(self) > float(o)

    def __eq__(self, o):
        return float(self) == float(o)

    def __hash__(self):
        return id(self)


class DummyLogger:
    def __init__(self, *a, **k):
        self.data: Dict[str, List] = {}
        self.checkpoint_fn = None

    def store(self, tab=None, **kwargs):
        for k, v in kwargs.items():
            self.data.setdefault(k if tab is None else f"{tab}/{k}", []).append(v)

    def get_mean(self, key: str) -> float:
        v = self.data.get(key, [])
        return sum(float(x) for x in v) / max(len(v), 1)

    def last(self, key: str) -> float:
        return float(self.data[key][-1])

    def reset(self):
        self.data = {}

    def write(self, step=None, display=False, **k):
        out = {k_: self.get_mean(k_) for k_ in self.data}
        self.reset()
        return out

    def write_without_reset(self, step=None):
        return {k_: self.get_mean(k_) for k_ in self.data}

    def save_config(self, *a, **k):
        pass

    def setup_checkpoint_fn(self, fn=None):
        self.checkpoint_fn = fn

    def save_checkpoint(self, suffix=None):
        pass


def store_stats(logger, st, mode: str, tab=None, keys=None) -> None:
    """Hand the statistics of the step just enqueued to ``logger.store``."""
    if logger is None:
        return
    kw = {} if tab is None else {"tab": tab}
    use = list(st.keys) if keys is None else list(keys)
    if mode == "sync":
        vals = st.read_stats()
        logger.store(**kw, **{k: vals[k] for k in use})
    elif mode == "lazy":
        step = st.host_step
        pend = getattr(st, "_pending", None)
        if pend is None:
            pend = st._pending = []
        vals = {k: LazyStat(st, step, k) for k in use}
        pend.extend(vals.values())
        # materialise before the device ring wraps (amortised: one sync per ring_len/2 steps)
        if len(pend) >= (st.ring_len // 2) * max(len(st.keys), 1):
            for v in pend:
                v.materialize()
            pend.clear()
        logger.store(**kw, **vals)
    elif mode != "none":
        raise ValueError(mode)
