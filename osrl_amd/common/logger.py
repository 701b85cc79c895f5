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
    """``max_keep``: entries kept per key; older ones are folded into a running (sum, count) so ``get_mean`` stays
    exact while a logger that is never ``write()``-n (the trainers' default) cannot grow without bound."""

    def __init__(self, *a, max_keep: int = 4096, **k):
        self.data: Dict[str, List] = {}
        self._folded: Dict[str, List[float]] = {}
        self.max_keep = int(max_keep)
        self.checkpoint_fn = None

    def store(self, tab=None, **kwargs):
        for k, v in kwargs.items():
            key = k if tab is None else f"{tab}/{k}"
            lst = self.data.setdefault(key, [])
            lst.append(v)
            if len(lst) > 2 * self.max_keep:  # fold the older half (materialises LazyStats that are still readable)
                old, self.data[key] = lst[:-self.max_keep], lst[-self.max_keep:]
                acc = self._folded.setdefault(key, [0.0, 0])
                for x in old:
                    try:
                        acc[0] += float(x)
                        acc[1] += 1
                    except RuntimeError:  # its ring slot was overwritten long ago: drop it from the mean
                        pass

    def get_mean(self, key: str) -> float:
        v = self.data.get(key, [])
        s, n = self._folded.get(key, (0.0, 0))
        return (s + sum(float(x) for x in v)) / max(n + len(v), 1)

    def last(self, key: str) -> float:
        return float(self.data[key][-1])

    def reset(self):
        self.data = {}
        self._folded = {}

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
        # materialise before the device ring wraps (amortised: one sync per ring_len/2 steps).  The trigger is the
        # AGE of the oldest pending step, not the entry count: with a `keys` subset the count grows slower than the ring
        if step - pend[0]._step >= st.ring_len // 2:
            todo = [v for v in pend if v._val is None]
            rows = st.read_stats_many({v._step for v in todo})  # ONE device->host copy for the whole backlog
            for v in todo:
                v._val = float(rows[v._step][st.index[v._key]])
                v._st = None
            pend.clear()
        logger.store(**kw, **vals)
    elif mode != "none":
        raise ValueError(mode)
