"""CPU oracle for the OSRL hot path (BC / CPQ / BCQ-Lag ``train_one_step``).

TEST INFRASTRUCTURE ONLY.  Nothing under ``osrl_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the CPU baseline.

It is a numpy restatement (hand-derived forward AND backward, explicit Adam,
explicit noise inputs) of the reference's PyTorch path.  Every function cites
the reference ``file:line`` it follows (paths relative to the reference root).
Parity is PINNED by ``tests/golden/*.npz`` -- vectors captured by importing the
reference itself (``tests/golden/make_golden.py``, torch 2.10 CPU) -- see
``tests/test_oracle_golden.py``.  The reference ships no tests of its own
(SURVEY.md section 4), so those goldens are the only pin there is.

State is a flat ``dict[str, np.ndarray]`` using the reference's ``state_dict``
key layout, so golden weights load without renaming.
Noise is always an explicit argument, in the draw order of SURVEY.md 8a-RNG.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

Array = np.ndarray
State = Dict[str, Array]

LOG_STD_MAX = 2.0  # osrl/common/net.py:148
LOG_STD_MIN = -20.0  # osrl/common/net.py:149


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def _act(name: str, x: Array) -> Array:
    if name == "relu":
        return np.maximum(x, 0)
    if name == "tanh":
        return np.tanh(x)
    if name == "id":
        return x
    raise ValueError(name)


def _act_grad_from_out(name: str, y: Array) -> Array:
    """d act / d pre-activation expressed with the activation OUTPUT y."""
    if name == "relu":
        return (y > 0).astype(y.dtype)  # threshold_backward: grad where out > 0
    if name == "tanh":
        return 1 - y * y
    if name == "id":
        return np.ones_like(y)
    raise ValueError(name)


class KinkBook:
    """Test bookkeeping for ReLU units that sit within fp32 round-off of their kink.

    ``threshold_backward`` passes gradient where the activation OUTPUT is > 0.  A unit whose pre-activation is within
    the rounding error of an fp32 dot product of zero can land on either side depending on the summation order, and
    then one row's contribution to one row of dW appears or vanishes.  While ``MLP.kink`` is set to a KinkBook,
    ``MLP.backward``
      (a) records for every relu layer the (row, unit) pairs with ``|pre| <= ulps * 2^-24 * (|x| @ |W|^T + |b|)`` under
          ``near[layer key]`` (one entry per backward call);
      (b) overrides relu' at the pairs listed in ``force[layer key] = (rows, units, values)`` -- a test can hand the
          oracle the decisions the device took (read off its saved activations) and demand agreement at the strict gate;
      (c) accumulates in ``allow[parameter key]`` an elementwise BOUND on what those undecidable units can change in
          that parameter's gradient: a flip at (r, j) moves dz[r, j] by |dy[r, j]|, hence row j of dW by |dy[r, j]| |x[r, :]|
          and db[j] by |dy[r, j]|; the uncertainty is carried down the same MLP (|U| @ |W| through each layer's
          derivative) so the rank-one change it makes to the layers below is bounded too.  A gradient element may then
          miss the strict gate only by its own ``allow`` -- not by a blanket budget.
    The uncertainty is NOT carried across networks (a critic's dX into the actor head): callers keep the absolute kink
    floor for that."""

    def __init__(self, ulps: float = 2.0):
        self.ulps = float(ulps)
        self.near: Dict[str, list] = {}
        self.force: Dict[str, tuple] = {}
        self.allow: Dict[str, Array] = {}

    def mask(self, key: str, x: Array, W: Array, b: Array, y: Array) -> Array:
        m = y > 0
        pre = x @ W.T + b
        bound = self.ulps * 2.0 ** -24 * (np.abs(x) @ np.abs(W).T + np.abs(b))
        rows, units = np.nonzero(np.abs(pre) <= bound)
        self.near.setdefault(key, []).append((rows, units))
        f = self.force.get(key)
        if f is not None:
            m = m.copy()
            m[f[0], f[1]] = np.asarray(f[2], bool)
        return m.astype(y.dtype)

    def layer(self, key: str, act: str, x: Array, W: Array, b: Array, y: Array, dy: Array, U: Optional[Array]):
        """dz of one layer + the uncertainty of the next dy.  ``U``: bound on |delta dy| coming from the layers above."""
        if act == "relu":
            m = self.mask(key, x, W, b, y)
        else:
            m = _act_grad_from_out(act, y)
        dz = dy * m
        Uz = np.zeros_like(dz) if U is None else U * np.abs(m)
        if act == "relu":
            rows, units = self.near[key][-1]
            if len(rows):
                Uz[rows, units] = np.maximum(Uz[rows, units], 0) + np.abs(dy[rows, units]) + (0 if U is None else U[rows, units])
        if Uz.any():
            ax = np.abs(x)
            self.allow[key + ".weight"] = self.allow.get(key + ".weight", 0) + Uz.T @ ax
            self.allow[key + ".bias"] = self.allow.get(key + ".bias", 0) + Uz.sum(0)
            return dz, Uz @ np.abs(W)
        return dz, None


class MLP:
    """``mlp()`` of osrl/common/net.py:12-30: ``y = act(x @ W.T + b)`` per layer.

    ``keys`` is the list of state_dict prefixes (``"critic.q_nets.0.0"`` ...),
    ``acts`` the activation after each layer.
    """
    kink: Optional["KinkBook"] = None  # tests only (KinkBook)

    def __init__(self, keys: Sequence[str], acts: Sequence[str]):
        assert len(keys) == len(acts)
        self.keys = list(keys)
        self.acts = list(acts)

    def forward(self, p: State, x: Array) -> Tuple[Array, List[Array]]:
        cache = [x]
        for k, a in zip(self.keys, self.acts):
            x = _act(a, x @ p[k + ".weight"].T + p[k + ".bias"])
            cache.append(x)
        return x, cache

    def backward(self, p: State, cache: List[Array], dy: Array, grads: State,
                 need_dx: bool = True) -> Optional[Array]:
        """Accumulates dW/db into ``grads`` (+=) and returns dX (or None)."""
        U = None  # (KinkBook: bound on what ulp-close relu units above may change in dy)
        for li in range(len(self.keys) - 1, -1, -1):
            k, a = self.keys[li], self.acts[li]
            if MLP.kink is not None:
                dz, U = MLP.kink.layer(k, a, cache[li], p[k + ".weight"], p[k + ".bias"], cache[li + 1], dy,
                                       U if li < len(self.keys) - 1 else None)
            else:
                dz = dy * _act_grad_from_out(a, cache[li + 1])
            grads[k + ".weight"] = grads.get(k + ".weight", 0) + dz.T @ cache[li]
            grads[k + ".bias"] = grads.get(k + ".bias", 0) + dz.sum(0)
            if li == 0 and not need_dx:
                return None
            dy = dz @ p[k + ".weight"]
        return dy


class Adam:
    """torch.optim.Adam, defaults betas=(0.9,0.999), eps=1e-8, no weight decay
    (osrl/algorithms/cpq.py:232-238; numerics SURVEY.md 8a-NUM):
    m<-b1 m+(1-b1)g; v<-b2 v+(1-b2)g^2; p<-p-lr/(1-b1^t)*m/(sqrt(v)/sqrt(1-b2^t)+eps)."""

    def __init__(self, keys: Sequence[str], lr: float, b1=0.9, b2=0.999, eps=1e-8,
                 weight_decay: float = 0.0):
        self.keys, self.lr, self.b1, self.b2, self.eps = list(keys), lr, b1, b2, eps
        self.wd = weight_decay  # decoupled (AdamW), osrl/algorithms/cdt.py:321-326
        self.t = 0
        self.m: State = {}
        self.v: State = {}

    def step(self, p: State, grads: State, lr: Optional[float] = None) -> None:
        lr = self.lr if lr is None else lr
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k in self.keys:
            g = np.asarray(grads[k], dtype=p[k].dtype)
            if k not in self.m:
                self.m[k] = np.zeros_like(p[k])
                self.v[k] = np.zeros_like(p[k])
            if self.wd:
                p[k] *= p[k].dtype.type(1 - lr * self.wd)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            denom = np.sqrt(self.v[k]) / math.sqrt(bc2) + self.eps
            p[k] -= ((lr / bc1) * self.m[k] / denom).astype(p[k].dtype)


def soft_update(p: State, tgt_prefix: str, src_prefix: str, tau: float) -> None:
    """``_soft_update`` osrl/algorithms/cpq.py:107-113: tgt<-tau*src+(1-tau)*tgt."""
    for k in list(p.keys()):
        if k.startswith(src_prefix + "."):
            kt = tgt_prefix + k[len(src_prefix):]
            p[kt] = (tau * p[k] + (1 - tau) * p[kt]).astype(p[kt].dtype)


def _keys_with_prefix(p: State, prefix: str) -> List[str]:
    return [k for k in p if k.startswith(prefix + ".")]


def _n_layers(p: State, prefix: str) -> int:
    """Number of Linear layers in an ``mlp()`` Sequential under ``prefix``."""
    idx = {int(k[len(prefix) + 1:].split(".")[0]) for k in _keys_with_prefix(p, prefix)}
    return len(idx)


def _seq_mlp(p: State, prefix: str, hidden_act: str, out_act: str) -> MLP:
    n = _n_layers(p, prefix)
    keys = [f"{prefix}.{2 * i}" for i in range(n)]
    return MLP(keys, [hidden_act] * (n - 1) + [out_act])


def _q_prefixes(p: State, prefix: str, attr: str) -> List[str]:
    n = len({k.split(".")[2] for k in _keys_with_prefix(p, f"{prefix}.{attr}")})
    return [f"{prefix}.{attr}.{i}" for i in range(n)]


def q_forward(p: State, prefixes: Sequence[str], x: Array):
    """EnsembleQCritic.forward osrl/common/net.py:228-233 (squeezed [rows] per net)."""
    outs, caches = [], []
    for pre in prefixes:
        net = _seq_mlp(p, pre, "relu", "id")
        y, c = net.forward(p, x)
        outs.append(y[:, 0])
        caches.append((net, c))
    return outs, caches


# --------------------------------------------------------------------------- #
# squashed Gaussian actor  (osrl/common/net.py:152-205)
# --------------------------------------------------------------------------- #
class SquashedGaussianActor:
    def __init__(self, p: State, prefix: str = "actor"):
        n = _n_layers(p, prefix + ".net")
        self.trunk = MLP([f"{prefix}.net.{2 * i}" for i in range(n)], ["relu"] * n)
        self.prefix = prefix

    def forward(self, p: State, obs: Array, eps: Optional[Array]):
        """Returns dict(mu, std, ls_raw, u, a(tanh u), logp, cache).  eps=None -> deterministic."""
        h, cache = self.trunk.forward(p, obs)
        pre = self.prefix
        mu = h @ p[pre + ".mu_layer.weight"].T + p[pre + ".mu_layer.bias"]
        ls_raw = h @ p[pre + ".log_std_layer.weight"].T + p[pre + ".log_std_layer.bias"]
        ls = np.clip(ls_raw, LOG_STD_MIN, LOG_STD_MAX)
        std = np.exp(ls)
        u = mu if eps is None else mu + std * eps
        # net.py:191-193
        logp = (-((u - mu) ** 2) / (2 * std * std) - ls - 0.5 * math.log(2 * math.pi)).sum(-1)
        logp = logp - (2 * (math.log(2) - u - np.logaddexp(0, -2 * u))).sum(-1)
        return dict(mu=mu, std=std, ls_raw=ls_raw, u=u, a=np.tanh(u), logp=logp, cache=cache, h=h)

    def backward(self, p: State, fw: dict, du: Array, eps: Array, grads: State) -> None:
        """Back-prop d loss / d u (pre-tanh sample) into the actor parameters."""
        pre = self.prefix
        dmu = du
        dls = du * eps * fw["std"]
        dls_raw = dls * ((fw["ls_raw"] >= LOG_STD_MIN) & (fw["ls_raw"] <= LOG_STD_MAX))
        h = fw["h"]
        grads[pre + ".mu_layer.weight"] = dmu.T @ h
        grads[pre + ".mu_layer.bias"] = dmu.sum(0)
        grads[pre + ".log_std_layer.weight"] = dls_raw.T @ h
        grads[pre + ".log_std_layer.bias"] = dls_raw.sum(0)
        dh = dmu @ p[pre + ".mu_layer.weight"] + dls_raw @ p[pre + ".log_std_layer.weight"]
        self.trunk.backward(p, fw["cache"], dh, grads, need_dx=False)


# --------------------------------------------------------------------------- #
# VAE  (osrl/common/net.py:290-339)
# --------------------------------------------------------------------------- #
class VAE:
    def __init__(self, act_lim: float, prefix: str = "vae"):
        self.act_lim, self.pre = act_lim, prefix
        self.enc = MLP([prefix + ".e1", prefix + ".e2"], ["relu", "relu"])
        self.dec = MLP([prefix + ".d1", prefix + ".d2", prefix + ".d3"], ["relu", "relu", "tanh"])

    def encode(self, p: State, obs: Array, act: Array):
        h, cache = self.enc.forward(p, np.concatenate([obs, act], 1))
        mean = h @ p[self.pre + ".mean.weight"].T + p[self.pre + ".mean.bias"]
        ls_raw = h @ p[self.pre + ".log_std.weight"].T + p[self.pre + ".log_std.bias"]
        std = np.exp(np.clip(ls_raw, -4, 15))  # net.py:325-326
        return mean, std, ls_raw, h, cache

    def decode(self, p: State, obs: Array, z: Array):
        """net.py:332-339 with z given (the caller clamps/draws it)."""
        t, cache = self.dec.forward(p, np.concatenate([obs, z], 1))
        return self.act_lim * t, cache

    @staticmethod
    def kl_rows(mean: Array, std: Array) -> Array:
        """-0.5*(1+log(std^2)-mean^2-std^2), elementwise (cpq.py:128,181)."""
        return -0.5 * (1 + np.log(std ** 2) - mean ** 2 - std ** 2)

    def loss_and_grads(self, p: State, obs: Array, act: Array, eps: Array, beta: float):
        """``vae_loss`` cpq.py:125-135 == bcql.py:122-132.  Returns (loss, grads)."""
        B, ad = act.shape
        od = obs.shape[1]
        mean, std, ls_raw, h, ecache = self.encode(p, obs, act)
        z = mean + std * eps
        u, dcache = self.decode(p, obs, z)
        L = mean.shape[1]
        recon = ((u - act) ** 2).mean()
        kl = self.kl_rows(mean, std).mean()
        loss = recon + beta * kl
        grads: State = {}
        du = 2 * (u - act) / (B * ad)
        # decoder output is act_lim*tanh(.) ; MLP.backward handles tanh' from the cached tanh
        ddin = self.dec.backward(p, dcache, du * self.act_lim, grads, need_dx=True)
        dz = ddin[:, od:]
        dmean = dz + beta * mean / (B * L)
        dstd = dz * eps + beta * (std - 1 / std) / (B * L)
        dls_raw = dstd * std * ((ls_raw >= -4) & (ls_raw <= 15))
        pre = self.pre
        grads[pre + ".mean.weight"] = dmean.T @ h
        grads[pre + ".mean.bias"] = dmean.sum(0)
        grads[pre + ".log_std.weight"] = dls_raw.T @ h
        grads[pre + ".log_std.bias"] = dls_raw.sum(0)
        dh = dmean @ p[pre + ".mean.weight"] + dls_raw @ p[pre + ".log_std.weight"]
        self.enc.backward(p, ecache, dh, grads, need_dx=False)
        return loss, grads


def quantile_linear(x: Array, q: float):
    """torch.quantile(x, q) on the flattened input: sort, linear interpolation at
    q*(n-1) (cpq.py:183)."""
    s = np.sort(x.reshape(-1))
    pos = q * (s.size - 1)
    lo = int(math.floor(pos))
    hi = min(lo + 1, s.size - 1)
    w = s.dtype.type(pos - lo)
    return s[lo] + (s[hi] - s[lo]) * w


def _vae_keys() -> List[str]:
    return [f"vae.{n}.{s}" for n in ("e1", "e2", "mean", "log_std", "d1", "d2", "d3")
            for s in ("weight", "bias")]


# --------------------------------------------------------------------------- #
# BC  (osrl/algorithms/bc.py)
# --------------------------------------------------------------------------- #
class OracleBC:
    """BC.actor_loss bc.py:45-52 + BCTrainer.train_one_step bc.py:103-109."""

    def __init__(self, params: State, max_action: float, actor_lr: float = 1e-3,
                 dtype=np.float32):
        self.p = {k: np.array(v, dtype=dtype) for k, v in params.items()}
        self.max_action = max_action
        self.pi = _seq_mlp(self.p, "actor.pi", "relu", "tanh")  # net.py:77-85
        self.opt = Adam(_keys_with_prefix(self.p, "actor"), actor_lr)
        self.dtype = dtype

    def act(self, obs: Array) -> Array:
        y, _ = self.pi.forward(self.p, np.asarray(obs, self.dtype))
        return self.max_action * y

    def train_one_step(self, observations, actions) -> Dict[str, float]:
        obs = np.asarray(observations, self.dtype)
        act = np.asarray(actions, self.dtype)
        y, cache = self.pi.forward(self.p, obs)
        pred = self.max_action * y
        loss = ((pred - act) ** 2).mean()
        grads: State = {}
        self.pi.backward(self.p, cache, 2 * (pred - act) / pred.size * self.max_action, grads,
                         need_dx=False)
        self.opt.step(self.p, grads)
        return {"loss/actor_loss": float(loss)}


# --------------------------------------------------------------------------- #
# CPQ  (osrl/algorithms/cpq.py)
# --------------------------------------------------------------------------- #
class OracleCPQ:
    """CPQ + CPQTrainer.train_one_step (cpq.py:294-313).  ``noise`` keys, in the
    reference's RNG draw order:
    eps_vae[B,2ad], eps_next_c[B,ad], eps_next_cc[B,ad], eps_pi_unused[B,ad],
    eps_ood[N,B,ad], eps_vae_ood[N*B,2ad] (result-irrelevant), eps_actor[B,ad]."""

    def __init__(self, params: State, *, max_action: float, sample_action_num: int = 10,
                 gamma: float = 0.99, tau: float = 0.005, beta: float = 0.5,
                 qc_scalar: float = 1.5, cost_limit: float = 10, episode_len: int = 300,
                 actor_lr=1e-4, critic_lr=1e-3, alpha_lr=1e-4, vae_lr=1e-3,
                 dtype=np.float32):
        self.p = {k: np.array(v, dtype=dtype) for k, v in params.items()}
        p = self.p
        self.dtype = dtype
        self.max_action, self.N = max_action, sample_action_num
        self.gamma, self.tau, self.beta = gamma, tau, beta
        # cpq.py:102-105
        self.q_thres = cost_limit * (1 - gamma ** episode_len) / (1 - gamma) / episode_len
        self.qc_thres = qc_scalar * self.q_thres
        self.log_alpha = 0.0  # cpq.py:93
        self.alpha_lr = alpha_lr
        self.actor = SquashedGaussianActor(p, "actor")
        self.vae = VAE(max_action, "vae")
        self.q = _q_prefixes(p, "critic", "q_nets")
        self.qc = _q_prefixes(p, "cost_critic", "q_nets")
        self.q_old = _q_prefixes(p, "critic_old", "q_nets")
        self.qc_old = _q_prefixes(p, "cost_critic_old", "q_nets")
        self.opt_actor = Adam(_keys_with_prefix(p, "actor"), actor_lr)
        self.opt_critic = Adam(_keys_with_prefix(p, "critic"), critic_lr)
        self.opt_cost = Adam(_keys_with_prefix(p, "cost_critic"), critic_lr)
        self.opt_vae = Adam(_vae_keys(), vae_lr)

    # cpq.py:115-123
    def _actor_forward(self, obs, eps):
        fw = self.actor.forward(self.p, obs, eps)
        return fw["a"] * self.max_action, fw

    def act(self, obs: Array) -> Array:
        """CPQ.act deterministic (cpq.py:240-252) on a batch of observations."""
        a, _ = self._actor_forward(np.asarray(obs, self.dtype), None)
        return a

    def _q_loss_backward(self, prefixes, x, backup, opt):
        B = x.shape[0]
        qs, caches = q_forward(self.p, prefixes, x)
        loss = sum(((q - backup) ** 2).mean() for q in qs)  # net.py:240-242
        grads: State = {}
        for q, (net, cache) in zip(qs, caches):
            net.backward(self.p, cache, (2 * (q - backup) / B)[:, None], grads, need_dx=False)
        return loss, grads, qs

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done,
                       noise: Dict[str, Array]) -> Dict[str, float]:
        dt = self.dtype
        obs, nobs, act = (np.asarray(a, dt) for a in (observations, next_observations, actions))
        rew, cost, done = (np.asarray(a, dt) for a in (rewards, costs, done))
        nz = {k: np.asarray(v, dt) for k, v in noise.items()}
        p, g, N = self.p, self.gamma, self.N
        B, od = obs.shape
        ad = act.shape[1]
        stats: Dict[str, float] = {}

        # ---- vae_loss  cpq.py:125-135
        loss_vae, gr = self.vae.loss_and_grads(p, obs, act, nz["eps_vae"], self.beta)
        self.opt_vae.step(p, gr)
        stats["loss/loss_vae"] = float(loss_vae)

        # ---- critic_loss  cpq.py:137-153
        na, _ = self._actor_forward(nobs, nz["eps_next_c"])
        xn = np.concatenate([nobs, na], 1)
        q_t = np.min(np.stack(q_forward(p, self.q_old, xn)[0]), 0)
        qc_t = np.min(np.stack(q_forward(p, self.qc_old, xn)[0]), 0)
        backup = rew + g * (1 - done) * (qc_t <= self.q_thres) * q_t
        x = np.concatenate([obs, act], 1)
        loss_c, gr, _ = self._q_loss_backward(self.q, x, backup.astype(dt), self.opt_critic)
        self.opt_critic.step(p, gr)
        stats["loss/critic_loss"] = float(loss_c)

        # ---- cost_critic_loss  cpq.py:155-201
        na, _ = self._actor_forward(nobs, nz["eps_next_cc"])
        xn = np.concatenate([nobs, na], 1)
        qc_t = np.min(np.stack(q_forward(p, self.qc_old, xn)[0]), 0)
        backup = cost + g * qc_t  # no (1-done): cpq.py:161
        fw = self.actor.forward(p, obs, None)  # only the distribution is used (cpq.py:164)
        sampled = fw["mu"][None] + fw["std"][None] * nz["eps_ood"]  # pre-tanh, cpq.py:166
        sampled = sampled.reshape(N * B, ad)
        stacked = np.tile(obs[None], (N, 1, 1)).reshape(N * B, od)  # j*B+b, cpq.py:170-174
        qc_s = np.min(np.stack(q_forward(p, self.qc_old, np.concatenate([stacked, sampled], 1))[0]), 0)
        qc_s = qc_s.reshape(N, B)
        mean, std, _, _, _ = self.vae.encode(p, stacked, sampled)
        kl = self.vae.kl_rows(mean, std).mean(1).reshape(N, B)  # cpq.py:181-182
        quant = quantile_linear(kl, 0.75)
        qc_ood = ((kl >= quant) * qc_s).mean(0)
        loss_cc, gr, _ = self._q_loss_backward(self.qc, x, backup.astype(dt), self.opt_cost)
        loss_cc = loss_cc - math.exp(self.log_alpha) * (qc_ood.mean() - self.qc_thres)
        self.opt_cost.step(p, gr)
        # cpq.py:193-195
        self.log_alpha += self.alpha_lr * math.exp(self.log_alpha) * float(self.qc_thres - qc_ood.mean())
        self.log_alpha = float(np.clip(self.log_alpha, -5.0, 5.0))
        stats["loss/cost_critic_loss"] = float(loss_cc)
        stats["loss/alpha_value"] = math.exp(self.log_alpha)

        # ---- actor_loss  cpq.py:203-222
        eps = nz["eps_actor"]
        fw = self.actor.forward(p, obs, eps)
        a = fw["a"] * self.max_action
        xa = np.concatenate([obs, a], 1)
        qs, caches = q_forward(p, self.q, xa)
        qstack = np.stack(qs)
        amin = np.argmin(qstack, 0)  # torch.min(dim=0) routes grad to the arg-min net
        q_pi = qstack[amin, np.arange(B)]
        qc_pi = np.min(np.stack(q_forward(p, self.qc, xa)[0]), 0)
        mask = (qc_pi <= self.q_thres).astype(dt)
        loss_a = -(mask * q_pi).mean()
        da = np.zeros((B, ad), dt)
        scratch: State = {}
        for i, (net, cache) in enumerate(caches):
            dq = (-mask / B) * (amin == i)
            dx = net.backward(p, cache, dq[:, None].astype(dt), scratch, need_dx=True)
            da += dx[:, od:]
        du = da * self.max_action * (1 - fw["a"] ** 2)
        gr = {}
        self.actor.backward(p, fw, du, eps, gr)
        self.opt_actor.step(p, gr)
        stats["loss/actor_loss"] = float(loss_a)

        # ---- sync_weight  cpq.py:224-230
        soft_update(p, "critic_old", "critic", self.tau)
        soft_update(p, "cost_critic_old", "cost_critic", self.tau)
        soft_update(p, "actor_old", "actor", self.tau)
        return stats


# --------------------------------------------------------------------------- #
# BCQ-Lag  (osrl/algorithms/bcql.py)
# --------------------------------------------------------------------------- #
class PID:
    """LagrangianPIDController osrl/common/net.py:356-387."""

    def __init__(self, KP, KI, KD, thres):
        self.KP, self.KI, self.KD, self.thres = KP, KI, KD, thres
        self.error_old = 0.0
        self.error_integral = 0.0

    def control(self, qc: Array) -> float:
        e = float(np.mean(qc - self.thres))
        d = max(e - self.error_old, 0.0)
        self.error_integral = max(self.error_integral + e, 0.0)
        self.error_old = e
        return max(self.KP * max(e, 0.0) + self.KI * self.error_integral + self.KD * d, 0.0)


class OracleBCQL:
    """BCQL + BCQLTrainer.train_one_step (bcql.py:283-306).  ``noise`` keys in draw
    order: eps_vae[B,2ad], z_c[N*B,2ad], z_cc[N*B,2ad], z_actor[B,2ad] (raw normal
    draws; the +-0.5 clamp of net.py:334-335 is applied here)."""

    def __init__(self, params: State, *, max_action: float, sample_action_num: int = 10,
                 gamma=0.99, tau=0.005, phi=0.05, lmbda=0.75, beta=0.5,
                 PID_gains=(0.1, 0.003, 0.001), cost_limit=10, episode_len=300,
                 actor_lr=1e-3, critic_lr=1e-3, vae_lr=1e-3, dtype=np.float32):
        self.p = {k: np.array(v, dtype=dtype) for k, v in params.items()}
        p = self.p
        self.dtype = dtype
        self.max_action, self.N = max_action, sample_action_num
        self.gamma, self.tau, self.phi, self.lmbda, self.beta = gamma, tau, phi, lmbda, beta
        self.qc_thres = cost_limit * (1 - gamma ** episode_len) / (1 - gamma) / episode_len
        self.controller = PID(*PID_gains, self.qc_thres)
        self.vae = VAE(max_action, "vae")
        self.pi = _seq_mlp(p, "actor.pi", "tanh", "tanh")  # net.py:54-55 with nn.Tanh hidden
        self.pi_old = _seq_mlp(p, "actor_old.pi", "tanh", "tanh")
        self.nets = {n: _q_prefixes(p, n, "q1_nets") + _q_prefixes(p, n, "q2_nets")
                     for n in ("critic", "cost_critic", "critic_old", "cost_critic_old")}
        self.nq = {n: len(_q_prefixes(p, n, "q1_nets")) for n in self.nets}
        self.opt_actor = Adam(_keys_with_prefix(p, "actor"), actor_lr)
        self.opt_critic = Adam(_keys_with_prefix(p, "critic"), critic_lr)
        self.opt_cost = Adam(_keys_with_prefix(p, "cost_critic"), critic_lr)
        self.opt_vae = Adam(_vae_keys(), vae_lr)

    def _perturb(self, net: MLP, obs, act):
        """MLPGaussianPerturbationActor.forward net.py:59-62."""
        t, cache = net.forward(self.p, np.concatenate([obs, act], 1))
        pre = act + self.phi * self.max_action * t
        return np.clip(pre, -self.max_action, self.max_action), pre, cache

    def _targets(self, name_old, nobs, z):
        p, N = self.p, self.N
        B = nobs.shape[0]
        obs_n = np.repeat(nobs, N, 0)  # b*N+j  (bcql.py:138)
        dec, _ = self.vae.decode(p, obs_n, np.clip(z, -0.5, 0.5))
        a_t, _, _ = self._perturb(self.pi_old, obs_n, dec)
        qs, _ = q_forward(p, self.nets[name_old], np.concatenate([obs_n, a_t], 1))
        n1 = self.nq[name_old]
        q1 = np.min(np.stack(qs[:n1]), 0)
        q2 = np.min(np.stack(qs[n1:]), 0)
        q = self.lmbda * np.minimum(q1, q2) + (1 - self.lmbda) * np.maximum(q1, q2)
        return q.reshape(B, N).max(1)  # bcql.py:146

    def _critic_update(self, name, x, backup, opt):
        B = x.shape[0]
        qs, caches = q_forward(self.p, self.nets[name], x)
        loss = sum(((q - backup) ** 2).mean() for q in qs)  # bcql.py:149-150
        grads: State = {}
        for q, (net, cache) in zip(qs, caches):
            net.backward(self.p, cache, (2 * (q - backup) / B)[:, None], grads, need_dx=False)
        opt.step(self.p, grads)
        return float(loss)

    def act(self, obs: Array, z: Array) -> Array:
        obs = np.asarray(obs, self.dtype)
        dec, _ = self.vae.decode(self.p, obs, np.clip(np.asarray(z, self.dtype), -0.5, 0.5))
        return self._perturb(self.pi, obs, dec)[0]

    def train_one_step(self, observations, next_observations, actions, rewards, costs, done,
                       noise: Dict[str, Array]) -> Dict[str, float]:
        dt = self.dtype
        obs, nobs, act = (np.asarray(a, dt) for a in (observations, next_observations, actions))
        rew, cost, done = (np.asarray(a, dt) for a in (rewards, costs, done))
        nz = {k: np.asarray(v, dt) for k, v in noise.items()}
        p, g = self.p, self.gamma
        B, od = obs.shape
        ad = act.shape[1]
        stats: Dict[str, float] = {}

        loss_vae, gr = self.vae.loss_and_grads(p, obs, act, nz["eps_vae"], self.beta)
        self.opt_vae.step(p, gr)
        stats["loss/loss_vae"] = float(loss_vae)

        x = np.concatenate([obs, act], 1)
        backup = rew + g * (1 - done) * self._targets("critic_old", nobs, nz["z_c"])
        stats["loss/critic_loss"] = self._critic_update("critic", x, backup.astype(dt), self.opt_critic)
        backup = cost + g * self._targets("cost_critic_old", nobs, nz["z_cc"])  # bcql.py:172
        stats["loss/cost_critic_loss"] = self._critic_update("cost_critic", x, backup.astype(dt),
                                                             self.opt_cost)

        # ---- actor_loss  bcql.py:181-216
        dec, _ = self.vae.decode(p, obs, np.clip(nz["z_actor"], -0.5, 0.5))
        a, pre, pcache = self._perturb(self.pi, obs, dec)
        xa = np.concatenate([obs, a], 1)

        def minmin(name):
            qs, caches = q_forward(p, self.nets[name], xa)
            n1 = self.nq[name]
            s1, s2 = np.stack(qs[:n1]), np.stack(qs[n1:])
            i1, i2 = np.argmin(s1, 0), np.argmin(s2, 0)
            m1, m2 = s1[i1, np.arange(B)], s2[i2, np.arange(B)]
            # torch.min(a,b) backward: grad to the smaller, split 1/2 on exact ties
            w1 = np.where(m1 < m2, 1.0, np.where(m1 == m2, 0.5, 0.0)).astype(dt)
            sel = [w1 * (i1 == i) for i in range(n1)] + [(1 - w1) * (i2 == i) for i in range(len(qs) - n1)]
            return np.minimum(m1, m2), sel, caches

        q_pi, sel_q, caches_q = minmin("critic")
        qc_pi, sel_qc, caches_qc = minmin("cost_critic")
        mult = self.controller.control(qc_pi)
        qc_penalty = ((qc_pi - self.qc_thres) * mult).mean()
        loss_a = -q_pi.mean() + qc_penalty
        da = np.zeros((B, ad), dt)
        scratch: State = {}
        for sel, (net, cache) in zip(sel_q, caches_q):
            da += net.backward(p, cache, (-sel / B)[:, None].astype(dt), scratch, True)[:, od:]
        for sel, (net, cache) in zip(sel_qc, caches_qc):
            da += net.backward(p, cache, (sel * mult / B)[:, None].astype(dt), scratch, True)[:, od:]
        inside = (pre >= -self.max_action) & (pre <= self.max_action)
        gr = {}
        self.pi.backward(p, pcache, da * inside * self.phi * self.max_action, gr, need_dx=False)
        self.opt_actor.step(p, gr)
        stats["loss/actor_loss"] = float(loss_a)
        stats["loss/qc_penalty"] = float(qc_penalty)
        stats["loss/lagrangian"] = float(mult)

        soft_update(p, "critic_old", "critic", self.tau)
        soft_update(p, "cost_critic_old", "cost_critic", self.tau)
        soft_update(p, "actor_old", "actor", self.tau)
        return stats


# --------------------------------------------------------------------------- #
# minibatch sources (osrl/common/dataset.py) -- used to check the on-device samplers
# --------------------------------------------------------------------------- #
def prepare_sequence_sample(traj: Dict[str, Array], start_idx: int, seq_len: int, reward_scale: float = 1.0,
                            cost_scale: float = 1.0):
    """SequenceDataset.__prepare_sample (dataset.py:749-775): slice, scale, tail zero-pad, mask."""
    sl = slice(start_idx, start_idx + seq_len)
    states, actions = np.asarray(traj["observations"][sl]), np.asarray(traj["actions"][sl])
    returns = np.asarray(traj["returns"][sl]) * reward_scale
    cost_returns = np.asarray(traj["cost_returns"][sl]) * cost_scale
    costs = np.asarray(traj["costs"][sl])
    time_steps = np.arange(start_idx, start_idx + seq_len)
    episode_cost = traj["cost_returns"][0] * cost_scale
    n = states.shape[0]
    mask = np.hstack([np.ones(n), np.zeros(seq_len - n)])

    def pad(a):
        out = np.zeros((seq_len,) + a.shape[1:], a.dtype)
        out[:n] = a
        return out

    return pad(states), pad(actions), pad(returns), pad(cost_returns), time_steps, mask, episode_cost, pad(costs)


def transition_sample(data: Dict[str, Array], idx, reward_scale: float = 1.0, cost_scale: float = 1.0):
    """TransitionDataset.__prepare_sample (dataset.py:832-842) for an index array."""
    done = np.logical_or(np.asarray(data["terminals"]) == 1, np.asarray(data["timeouts"]) == 1).astype(np.float32)
    return (data["observations"][idx], data["next_observations"][idx], data["actions"][idx],
            data["rewards"][idx] * reward_scale, data["costs"][idx] * cost_scale, done[idx])


def rollout(policy, env, episode_len: int, cost_scale: float = 1.0):
    """XTrainer.rollout (cpq.py:330-347): policy(obs) -> action; returns (ret, len, cost)."""
    obs, info = env.reset()
    ret, cost, n = 0.0, 0.0, 0
    for _ in range(episode_len):
        obs, reward, terminated, truncated, info = env.step(policy(obs))
        ret += reward
        cost += info["cost"] * cost_scale
        n += 1
        if terminated or truncated:
            break
    return ret, n, cost
