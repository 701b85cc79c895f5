"""Host-side plumbing shared by the BC / CPQ / BCQ-Lag step engines.

* ``FlatGroup``  -- one optimizer group as flat fp32 HBM buffers (param, Adam m, Adam v, optional
  Polyak target, split-K gradient slabs).  ``nn.Parameter``s of the model are VIEWS into ``p`` /
  ``tgt`` so the reference's ``state_dict`` layout is preserved while the optimizer + Polyak update
  is ONE streaming kernel per group (csrc/optim.hip).
* ``NetDesc``    -- pointers/dims of an ensemble of identical MLPs (``osrl_mlp_t``).
* ``MlpRun``     -- activation / gradient buffers of one (NetDesc, rows) use + fwd / bwd launches.
* ``DwPlan``     -- the static device-resident work list of the weight-gradient GEMM launch.

PyTorch is used for device memory and streams only; all arithmetic is in libosrl_amd.so.
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import _lib as L
from . import plan as _plan


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def cur_stream() -> int:
    """hipStream_t of torch's current stream on the current device (as an int for ctypes)."""
    if not torch.cuda.is_available():
        raise RuntimeError("osrl_amd: kernels need a HIP device (no CPU fallback)")
    if _raw_stream is not None:  # ~0.2 us instead of ~3 us for building a torch.cuda.Stream object per launch
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


LAYOUT_ONLY_OK = False  # tests flip this to build parameter layouts on a GPU-less host (no launches possible)


def require_cuda(device) -> torch.device:
    dev = torch.device(device)
    if LAYOUT_ONLY_OK and dev.type == "cpu":
        return dev
    if dev.type != "cuda":
        raise RuntimeError(
            f"osrl_amd runs on MI355X only (device={device!r}); there is no CPU fallback. "
            "Use device='cuda' / 'cuda:N'.")
    if not torch.cuda.is_available():
        raise RuntimeError("osrl_amd: no HIP device visible (torch.cuda.is_available() is False)")
    return dev


def _align4(n: int) -> int:
    return (n + 3) & ~3


class FlatGroup:
    """Flat storage of one optimizer group.  Usage: add()* -> finalize() -> view()/tgt_view()."""

    def __init__(self, name: str, device, with_target: bool = False):
        self.name = name
        self.device = torch.device(device)
        self.with_target = with_target
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self._n = 0
        self.p = self.m = self.v = self.tgt = self.slabs = None
        self.n_splits = 0
        self.weights: List[str] = []          # keys of [out,in] weights that get packed copies
        self.pf = self.pb = self.tf = None    # packed fwd / packed W^T of p, packed fwd of tgt
        self.f_off: Dict[str, int] = {}
        self.b_off: Dict[str, int] = {}
        self.aliases: set = set()
        self._retired_slabs: list = []        # see ensure_slabs
        # bumped whenever another plan is attached to slabs that an earlier engine's plans (and captured graphs) use: that
        # engine's "rows beyond my split count stay zero" no longer holds once the newcomer writes there, so engines
        # compare epochs before every replay and re-capture / refuse (stale_plans())
        self.slab_epoch = 0

    def add(self, key: str, shape: Sequence[int], align: bool = True) -> int:
        assert self.p is None, "FlatGroup already finalized"
        if align:
            self._n = _align4(self._n)
        off = self._n
        n = 1
        for s in shape:
            n *= int(s)
        self.layout[key] = (off, tuple(int(s) for s in shape))
        self._n += n
        return off

    def mark_weight(self, key: str) -> None:
        """Register a 2-D [out,in] tensor (or packed-head alias) whose fragment-ordered copies the
        MLP kernels read (csrc/mlp.hip pack_kernel)."""
        assert len(self.layout[key][1]) == 2
        if key not in self.weights:
            self.weights.append(key)

    def finalize(self) -> None:
        n = max(_align4(self._n), 4)
        self.n = n
        z = lambda k=n: torch.zeros(k, dtype=torch.float32, device=self.device)  # noqa: E731
        self.p, self.m, self.v = z(), z(), z()
        self.tgt = z() if self.with_target else None
        if self.weights:
            r16 = lambda x: (x + 15) // 16 * 16  # noqa: E731
            nf = nb = 0
            ents = (L.PackEntryT * len(self.weights))()
            self._max_pack = 0
            for i, k in enumerate(self.weights):
                off, (o, ii) = self.layout[k]
                self.f_off[k], self.b_off[k] = nf, nb
                ents[i].src_off, ents[i].f_off, ents[i].b_off, ents[i].out, ents[i].in_ = off, nf, nb, o, ii
                fs, bs = r16(ii) * r16(o), r16(o) * (r16(ii) + 16)
                self._max_pack = max(self._max_pack, fs, bs)
                nf += fs
                nb += bs
            self.pf, self.pb = z(nf), z(nb)
            self.tf = z(nf) if self.with_target else None
            self._ents = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(self.device)
            # flat parameter -> position in the packed copies, for the optimizer step's fused refresh
            import numpy as np
            mf, mb = np.full(n, -1, np.int32), np.full(n, -1, np.int32)
            for k in self.weights:
                off, (o, ii) = self.layout[k]
                r, c = np.meshgrid(np.arange(o), np.arange(ii), indexing="ij")
                flat = (off + r * ii + c).ravel()
                assert (mf[flat] < 0).all(), f"{k}: parameter packed twice"
                mf[flat] = (self.f_off[k] + ((c // 4) * r16(o) + r) * 4 + c % 4).ravel()
                mb[flat] = (self.b_off[k] + ((r // 4) * (r16(ii) + 16) + c) * 4 + r % 4).ravel()
            self._map_f = torch.from_numpy(mf).to(self.device)
            self._map_b = torch.from_numpy(mb).to(self.device)

    def repack(self, params: bool = True, target: bool = True) -> None:
        """Refresh the packed weight copies from the canonical buffers (async, current stream)."""
        if not self.weights:
            return
        lib = L.load()
        if params:
            L.check(lib.osrl_pack_weights(self.p.data_ptr(), self.pf.data_ptr(), self.pb.data_ptr(),
                                          self._ents.data_ptr(), len(self.weights), self._max_pack, cur_stream()),
                    "osrl_pack_weights")
        if target and self.tgt is not None:
            L.check(lib.osrl_pack_weights(self.tgt.data_ptr(), self.tf.data_ptr(), None, self._ents.data_ptr(),
                                          len(self.weights), self._max_pack, cur_stream()), "osrl_pack_weights")

    def ensure_slabs(self, n_splits: int) -> None:
        """Grow-only.  A hipGraph captured by an earlier engine on this group holds the RAW address of the slab
        tensor it was captured with (dW writes, slab reduction and Adam reads all use that one address, so the graph
        stays self-consistent); the replaced tensor is therefore kept alive for the group's lifetime instead of
        going back to the caching allocator, where a replay would read and write freed memory.

        A retained tensor is ZEROED when another plan is attached to it: the dW kernels STORE rows [0, own splits) of
        their own parameter ranges and the consumer sums ``cur_splits`` rows of everything, so a row that the new
        plans never write must not keep what an earlier engine (bigger batch => more splits, or a different mix of
        split counts inside one group: CDT's token / per-sample / prefix plans) left there."""
        if self.slabs is None or self.n_splits < n_splits:
            if self.slabs is not None:
                self._retired_slabs.append(self.slabs)
                self.slab_epoch += 1
            self.slabs = torch.zeros(n_splits, self.n, dtype=torch.float32, device=self.device)
            self.n_splits = n_splits
        else:
            self.slabs.zero_()
            self.slab_epoch += 1

    def offset(self, key: str) -> int:
        return self.layout[key][0]

    def alias(self, key: str, first_key: str, shape: Sequence[int]) -> None:
        """Name a contiguous range that spans several adjacent tensors (packed mu|log_std heads)."""
        self.layout[key] = (self.layout[first_key][0], tuple(int(s) for s in shape))
        self.aliases.add(key)

    # ---- optimizer state by parameter key (checkpoint / resume; common/checkpoint.py) ----
    def optim_state(self) -> Dict[str, Dict[str, torch.Tensor]]:
        """Adam moments per parameter key, in ``torch.optim.Adam``'s vocabulary (host copies)."""
        keys = [k for k in self.layout if k not in self.aliases]
        return {"exp_avg": {k: self._view(self.m, k).cpu().clone() for k in keys},
                "exp_avg_sq": {k: self._view(self.v, k).cpu().clone() for k in keys}}

    def load_optim_state(self, state: Dict[str, Dict[str, torch.Tensor]]) -> None:
        keys = [k for k in self.layout if k not in self.aliases]
        for name, buf in (("exp_avg", self.m), ("exp_avg_sq", self.v)):
            missing = [k for k in keys if k not in state[name]]
            extra = [k for k in state[name] if k not in keys]
            if missing or extra:
                raise KeyError(f"optimizer state of group {self.name!r}: missing {missing}, unexpected {extra}")
            for k in keys:
                dst = self._view(buf, k)
                src = torch.as_tensor(state[name][k], dtype=torch.float32)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"{name}[{k}]: shape {tuple(src.shape)} != {tuple(dst.shape)}")
                dst.copy_(src)

    def _view(self, buf: torch.Tensor, key: str) -> torch.Tensor:
        off, shape = self.layout[key]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)

    def view(self, key: str) -> torch.Tensor:
        return self._view(self.p, key)

    def tgt_view(self, key: str) -> torch.Tensor:
        return self._view(self.tgt, key)

    def grad_view(self, key: str, reduce: bool = True) -> torch.Tensor:
        """Summed gradient of one tensor (debug / tests)."""
        off, shape = self.layout[key]
        n = 1
        for s in shape:
            n *= s
        g = self.slabs[:self.cur_splits, off:off + n]
        return g.sum(0).view(shape) if reduce else g

    cur_splits = 1

    def adam_step(self, lr: float, st_ptr: int, tau: float = 0.0, betas=(0.9, 0.999), eps=1e-8,
                  weight_decay: float = 0.0, gscale: Optional[torch.Tensor] = None,
                  polyak: bool = True) -> None:
        lib = L.load()
        tgt = _ptr(self.tgt) if (polyak and self.tgt is not None) else None
        if self.weights:  # parameters, Polyak targets and their packed copies in one pass
            L.check(lib.osrl_adam_step_packed(self.p.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), tgt,
                                              self.slabs.data_ptr(), self.cur_splits, self.n, self.n, lr, betas[0],
                                              betas[1], eps, weight_decay, tau, _ptr(gscale), st_ptr,
                                              self._map_f.data_ptr(), self._map_b.data_ptr(), self.pf.data_ptr(),
                                              self.pb.data_ptr(), _ptr(self.tf) if tgt is not None else None,
                                              cur_stream()), "osrl_adam_step_packed")
        else:
            L.check(lib.osrl_adam_step(self.p.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), tgt,
                                       self.slabs.data_ptr(), self.cur_splits, self.n, self.n, lr, betas[0],
                                       betas[1], eps, weight_decay, tau, _ptr(gscale), st_ptr, cur_stream()),
                    "osrl_adam_step")


    def polyak_step(self, tau: float) -> None:
        """The target update alone (after ``adam_step(..., polyak=False)``): same bits as the step carried by Adam."""
        assert self.tgt is not None
        L.check(L.load().osrl_polyak(self.p.data_ptr(), self.tgt.data_ptr(), self.n, tau,
                                     self._map_f.data_ptr() if self.weights else None,
                                     _ptr(self.tf) if self.weights else None, cur_stream()), "osrl_polyak")


def slab_epochs(model) -> Tuple[int, ...]:
    """The slab epochs of a model's optimizer groups, as an engine records them when its plans are built."""
    return tuple(g.slab_epoch for g in model.groups.values())


def check_plans_current(engine) -> None:
    """Called by the step engines before a step: an engine built EARLIER on the same model whose groups have since had
    another engine's plans attached must not run any more -- its dW plans' split counts no longer describe which slab
    rows are written (FlatGroup.ensure_slabs) and a captured graph of it would sum rows the newcomer fills.
    ``model.engine(...)`` always hands out the newest engine; holding on to an old one is the misuse this catches."""
    cur = slab_epochs(engine.model)
    if getattr(engine, "_slab_epochs", None) is None:  # (every engine records them at the end of its __init__)
        raise RuntimeError("osrl_amd: step engine without recorded slab epochs (engine __init__ must set _slab_epochs)")
    if cur != engine._slab_epochs:
        raise RuntimeError("osrl_amd: this step engine is stale -- another engine was built on the same model after it "
                           "(different batch size / plan); use the engine model.engine(...) returns now")


class LayerRef:
    """One Linear layer of a net: canonical weight/bias tensors (views into ``group``) + the key of its
    packed copies.  ``target=True`` reads the Polyak-target copy of the group."""
    __slots__ = ("W", "b", "group", "key", "bkey", "target", "wparams", "bparams")

    def __init__(self, W: torch.Tensor, b: torch.Tensor, group: FlatGroup, key: str, bkey: str,
                 target: bool = False, wparams=None, bparams=None):
        self.W, self.b, self.group, self.key, self.bkey, self.target = W, b, group, key, bkey, target
        # the nn.Parameters behind W / b (stacked along dim 0 for packed heads) -- autograd routing in ops.py
        self.wparams, self.bparams = wparams, bparams

    @property
    def wf_ptr(self) -> int:
        g = self.group
        return (g.tf if self.target else g.pf).data_ptr() + 4 * g.f_off[self.key]

    @property
    def wb_ptr(self) -> Optional[int]:
        g = self.group
        return None if self.target else g.pb.data_ptr() + 4 * g.b_off[self.key]


class NetDesc:
    """An ensemble of identical MLPs: ``nets[e][l]`` = LayerRef."""

    def __init__(self, nets: Sequence[Sequence[LayerRef]], acts: Sequence[str], out_scale: float = 1.0):
        E, nl = len(nets), len(nets[0])
        if not (1 <= E <= L.MAX_NETS) or not (1 <= nl <= L.MAX_LAYERS):
            raise ValueError(f"fused MLP supports <= {L.MAX_NETS} nets and <= {L.MAX_LAYERS} Linear layers "
                             f"(got {E} nets, {nl} layers)")
        self.E, self.nl = E, nl
        dims = [nets[0][0].W.shape[1]] + [nets[0][l].W.shape[0] for l in range(nl)]
        if max(dims) > L.MAX_WIDTH:
            raise ValueError(f"layer width {max(dims)} > {L.MAX_WIDTH} unsupported by the fused MLP kernels")
        self.dims = dims
        self.acts = [L.ACT_CODES[a] for a in acts]
        self.out_scale = float(out_scale)
        self.nets = nets
        self.keys = [[(r.key, r.bkey) for r in net] for net in nets]
        d = L.MlpT()
        d.n_layers, d.n_nets = nl, E
        for i, v in enumerate(dims):
            d.dims[i] = v
        for i, a in enumerate(self.acts):
            d.acts[i] = a
        d.out_scale = self.out_scale
        for e in range(E):
            for l in range(nl):
                r = nets[e][l]
                assert r.b.is_contiguous() and r.W.dtype == torch.float32
                assert tuple(r.W.shape) == (dims[l + 1], dims[l]) and tuple(r.b.shape) == (dims[l + 1],)
                assert r.key in r.group.f_off, f"{r.key} has no packed copy (FlatGroup.mark_weight)"
                d.Wf[e][l] = r.wf_ptr
                d.Wb[e][l] = r.wb_ptr
                d.b[e][l] = r.b.data_ptr()
        self.c = d

    def groups(self):
        seen = []
        for net in self.nets:
            for r in net:
                if r.group not in seen:
                    seen.append(r.group)
        return seen

    def subset(self, idx: Sequence[int]) -> "NetDesc":
        d = NetDesc([self.nets[i] for i in idx], [_ACT_NAMES[a] for a in self.acts], self.out_scale)
        d.c.tile_rows = self.c.tile_rows
        return d


_ACT_NAMES = {L.ACT_ID: "id", L.ACT_RELU: "relu", L.ACT_TANH: "tanh"}


def concat_nets(a: NetDesc, b: NetDesc) -> NetDesc:
    """One launch over two ensembles of identical shape (e.g. critic_old + cost_critic_old)."""
    assert a.dims == b.dims and a.acts == b.acts and a.out_scale == b.out_scale
    return NetDesc(list(a.nets) + list(b.nets), [_ACT_NAMES[x] for x in a.acts], a.out_scale)


class MlpRun:
    """Buffers + launches for one use of a NetDesc on ``rows`` rows.

    ``save=True`` keeps every layer's activations (+ the concatenated input) for the backward pass;
    otherwise only the net outputs ``y`` [E, rows, out] are written.
    """

    def __init__(self, net: NetDesc, rows: int, save: bool, device, save_nets: Optional[Sequence[int]] = None,
                 wg_cap: int = 0, tile_rows: int = 0):
        self.net, self.rows, self.save = net, rows, save
        # forward descriptor of THIS use: a capped launch walks its tiles with at most wg_cap workgroups in flight;
        # tile_rows picks the forward row tile (80 = one 4-wave workgroup per CU, csrc/mlp.hip waves_per_simd)
        self.fwd_c = net.c
        if wg_cap or tile_rows:
            self.fwd_c = L.MlpT.from_buffer_copy(net.c)
            self.fwd_c.wg_cap = int(wg_cap)
            if tile_rows:
                self.fwd_c.tile_rows = int(tile_rows)
        f = dict(dtype=torch.float32, device=device)
        E, nl, dims = net.E, net.nl, net.dims
        self.y = torch.zeros(E, rows, dims[-1], **f)
        self.acts_c = L.ActsT()
        self.h: List[List[Optional[torch.Tensor]]] = [[None] * nl for _ in range(E)]
        self.x = None
        sn = set(range(E) if save_nets is None else save_nets) if save else set()
        if save:
            self.x = torch.zeros(rows, dims[0], **f)
            self.acts_c.x = self.x.data_ptr()
        for e in range(E):
            for l in range(nl - 1):
                if e in sn:
                    self.h[e][l] = torch.zeros(rows, dims[l + 1], **f)
                    self.acts_c.h[e][l] = self.h[e][l].data_ptr()
            self.h[e][nl - 1] = self.y[e]
            self.acts_c.h[e][nl - 1] = self.y[e].data_ptr()
        self.saved_nets = sorted(sn)
        self.dz: Optional[List[List[torch.Tensor]]] = None
        self.dx = None
        self.grads_c = None
        self.bwd_net = None

    # ---- forward ----
    def _rows(self, src0: torch.Tensor, src1: Optional[torch.Tensor] = None, map0=L.MAP_ID, div0=1,
              map1=L.MAP_ID, div1=1):
        r = L.RowsT()
        r.rows = self.rows
        r.d0, r.map0, r.div0 = src0.shape[-1], map0, div0
        r.src0 = src0.data_ptr()
        if src1 is not None:
            r.d1, r.map1, r.div1 = src1.shape[-1], map1, div1
            r.src1 = src1.data_ptr()
        else:
            r.d1, r.map1, r.div1 = 0, L.MAP_ID, 1
        assert r.d0 + r.d1 == self.net.dims[0], (r.d0, r.d1, self.net.dims)
        return r

    def forward_with(self, args, other: "MlpRun", other_args, tail: Optional["L.TailT"] = None,
                     other_tail: Optional["L.TailT"] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """This forward and an independent one of ``other`` in ONE launch (osrl_mlp_forward2);
        ``args`` / ``other_args`` are the positional arguments of the two ``forward`` calls; ``tail`` / ``other_tail``:
        a forward tail (osrl_mlp_tail_t) per problem, applied to its net 0's output tile by the same launch."""
        r0, r1 = self._rows(*args), other._rows(*other_args)
        if tail is not None or other_tail is not None:
            L.check(L.load().osrl_mlp_forward2_tail(
                C.byref(self.net.c), C.byref(r0), C.byref(self.acts_c), None if tail is None else C.byref(tail),
                C.byref(other.net.c), C.byref(r1), C.byref(other.acts_c),
                None if other_tail is None else C.byref(other_tail), cur_stream()), "osrl_mlp_forward2_tail")
            return self.y, other.y
        L.check(L.load().osrl_mlp_forward2(C.byref(self.net.c), C.byref(r0), C.byref(self.acts_c),
                                           C.byref(other.net.c), C.byref(r1), C.byref(other.acts_c), cur_stream()),
                "osrl_mlp_forward2")
        return self.y, other.y

    def share_k16(self, d0: int, src0_rows: int, copies: int) -> int:
        """How many whole 16-column k-steps of layer 0 a shared-src0-rows launch (osrl_rows_t.share0) can run once per src0
        row: those inside src0's ``d0`` columns, at least one k-step left -- or 0 where the form does not exist (tile shape:
        src0 rows a multiple of 16, copies a multiple of the 80-row tile's five row blocks)."""
        dims = self.net.dims
        k16 = d0 // 16
        if k16 * 16 >= ((dims[0] + 15) // 16) * 16:
            k16 -= 1
        if k16 < 1 or src0_rows % 16 or copies % 5 or self.rows != src0_rows * copies:
            return 0
        return k16

    def forward(self, src0: torch.Tensor, src1: Optional[torch.Tensor] = None, map0=L.MAP_ID, div0=1,
                map1=L.MAP_ID, div1=1, tail: Optional["L.TailT"] = None, row_list: Optional[torch.Tensor] = None,
                n_rows_dev: Optional[torch.Tensor] = None, share_k16: int = 0) -> torch.Tensor:
        """``tail``: an osrl_mlp_tail_t applied by the launch itself to net 0's output tile (osrl_mlp_forward_tail).
        ``row_list`` / ``n_rows_dev`` (int32 device tensors): the launch runs on the rows ``row_list[0 .. n_rows_dev[0])``
        of the virtual input, outputs compacted in that order (osrl_rows_t.row_list; ``self.rows`` = the capacity).
        ``share_k16`` > 0 (rows = n * div0 + b, map0 = MAP_MOD): tiles of shared src0 rows, the first ``16 share_k16`` input
        columns of layer 0 once per src0 row of a tile (osrl_rows_t.share0; ``self.share_k16()`` says how many)."""
        r = L.RowsT()
        r.rows = self.rows
        if share_k16:
            r.share0, r.share_k16 = 1, int(share_k16)
        if row_list is not None:
            assert row_list.dtype == torch.int32 and n_rows_dev.dtype == torch.int32 and row_list.numel() >= self.rows
            r.row_list, r.n_rows_dev = row_list.data_ptr(), n_rows_dev.data_ptr()
        r.d0, r.map0, r.div0 = src0.shape[-1], map0, div0
        r.src0 = src0.data_ptr()
        if src1 is not None:
            r.d1, r.map1, r.div1 = src1.shape[-1], map1, div1
            r.src1 = src1.data_ptr()
        else:
            r.d1, r.map1, r.div1 = 0, L.MAP_ID, 1
        assert r.d0 + r.d1 == self.net.dims[0], (r.d0, r.d1, self.net.dims)
        if tail is not None:
            L.check(L.load().osrl_mlp_forward_tail(C.byref(self.fwd_c), C.byref(r), C.byref(self.acts_c), C.byref(tail),
                                                   cur_stream()), "osrl_mlp_forward_tail")
            return self.y
        L.check(L.load().osrl_mlp_forward(C.byref(self.fwd_c), C.byref(r), C.byref(self.acts_c), cur_stream()),
                "osrl_mlp_forward")
        return self.y

    # ---- backward ----
    def setup_backward(self, dy: torch.Tensor, need_dz: bool = True, dx_cols: Optional[Tuple[int, int]] = None,
                       dx_out: Optional[torch.Tensor] = None) -> None:
        """dy: [len(saved_nets), rows, out] gradient wrt the outputs of the saved nets."""
        assert self.save
        idx = self.saved_nets
        net = self.net if len(idx) == self.net.E else self.net.subset(idx)
        self.bwd_net = net
        f = dict(dtype=torch.float32, device=dy.device)
        g = L.GradsT()
        sv = L.ActsT()
        sv.x = self.x.data_ptr()
        self.dy = dy
        self.dz = []
        for j, e in enumerate(idx):
            g.dy[j] = dy[j].data_ptr()
            row = []
            for l in range(net.nl):
                sv.h[j][l] = self.h[e][l].data_ptr()
                if need_dz:
                    t = torch.zeros(self.rows, net.dims[l + 1], **f)
                    g.dz[j][l] = t.data_ptr()
                    row.append(t)
            self.dz.append(row)
        if dx_cols is not None:
            c0, nc = dx_cols
            self.dx = dx_out if dx_out is not None else torch.zeros(len(idx), self.rows, nc, **f)
            assert tuple(self.dx.shape) == (len(idx), self.rows, nc) and self.dx.is_contiguous()
            for j in range(len(idx)):
                g.dx[j] = self.dx[j].data_ptr()
            g.dx_col0, g.dx_cols = c0, nc
        self.grads_c, self.saved_c = g, sv

    def backward_dz(self, tail: Optional["L.TailT"] = None, seed: Optional["L.SeedT"] = None) -> None:
        """``tail``: an osrl_mlp_tail_t applied by the launch itself to net 0's dX slice (osrl_mlp_backward_dz_tail);
        ``seed``: an osrl_mlp_seed_t -- the launch computes dL/d(output) itself (osrl_mlp_backward_dz_seed)."""
        if seed is not None:
            L.check(L.load().osrl_mlp_backward_dz_seed(C.byref(self.bwd_net.c), self.rows, C.byref(self.saved_c),
                                                       C.byref(self.grads_c), None if tail is None else C.byref(tail),
                                                       C.byref(seed), cur_stream()), "osrl_mlp_backward_dz_seed")
            return
        if tail is not None:
            L.check(L.load().osrl_mlp_backward_dz_tail(C.byref(self.bwd_net.c), self.rows, C.byref(self.saved_c),
                                                       C.byref(self.grads_c), C.byref(tail), cur_stream()),
                    "osrl_mlp_backward_dz_tail")
            return
        L.check(L.load().osrl_mlp_backward_dz(C.byref(self.bwd_net.c), self.rows, C.byref(self.saved_c),
                                              C.byref(self.grads_c), cur_stream()), "osrl_mlp_backward_dz")

    def dw_entries(self) -> List[Tuple[torch.Tensor, torch.Tensor, str, str]]:
        """(dz, input activation, weight key, bias key) for every layer of every saved net."""
        out = []
        net = self.bwd_net
        for j, e in enumerate(self.saved_nets):
            for l in range(net.nl):
                a = self.x if l == 0 else self.h[e][l - 1]
                wk, bk = net.keys[j][l]
                out.append((self.dz[j][l], a, wk, bk))
        return out


# dW + the group's optimizer step in one launch (osrl_mlp_backward_dw_tiles_adam; single device, weight decay 0, no clip
# scale).  Measured on the CPQ step (round 4, gpurun_out/r4a, DESIGN_LOG.md): with row splits the tile's slabs are
# exchanged between workgroups INSIDE the launch -- store, acknowledge, count, read back: four dependent device-coherent
# round trips, 9-14 us behind the dW k-loop against the 6.4-10 us of the separate optimizer launch (actor group 23.0 vs
# 11.5 + 6.6 us isolated, critic 41.9 vs 20.5 + 8.0; C2 1935 vs 2230 steps/s) -- so "auto" fuses only plans whose tiles
# have ONE split (the gradient is then complete in LDS: no slab, no exchange; small batches).  "1" forces it for every
# flat plan (the bit-equality test, A/B runs), "0" never fuses.
FUSE_DW_ADAM = _plan.knob("OSRL_FUSE_DW_ADAM", "auto", "dW + optimizer step in one launch: 1 / 0 / auto (one-split plans)", operator=True)


class DwPlan:
    """Static work list for osrl_mlp_backward_dw over one optimizer group."""

    BIG_ROWS = 8192  # from this many rows on, fully 128x128-tiled layers take the one-wave-per-tile kernel
    COOP = _plan.knob("OSRL_DW_COOP", "1", "token-matrix dW on 256 x 256 workgroup tiles") == "1"  # ... and fully 256x256-tiled ones the workgroup-per-tile kernel
    FLAT_DEFAULT = _plan.knob("OSRL_DW_FLAT", "1", "dW on the flat (tile, split) work list") == "1"

    def __init__(self, group: FlatGroup, entries: Sequence[Tuple[torch.Tensor, torch.Tensor, str, str]],
                 rows: int, device, n_splits: Optional[int] = None, big: Optional[bool] = None,
                 tile_blocks: int = 0):
        """``tile_blocks`` = 5: the flat (tile, row split) work list on 80 x 80 tiles (osrl_mlp_backward_dw_tiles) --
        for groups of 400-wide layers (25 column blocks = 5 x 5: no ragged tiles); a tile gets row splits in
        proportion to its blocks, so that every workgroup carries about the same number of MFMAs."""
        self.group, self.rows = group, rows
        self.tile_blocks = int(tile_blocks)
        use_big = (rows >= self.BIG_ROWS) if big is None else bool(big)
        if not self.tile_blocks and self.FLAT_DEFAULT and not use_big:
            # round 3: the training-row groups take the flat-list kernel too (64 x 64 tiles, the split policy of
            # mlp_dw_kernel below): its unmasked, counted-wait k-loop runs a step's MFMAs under the next step's loads
            self.tile_blocks = 4
            if n_splits is None:
                n_t = sum(((group.layout[e[2]][1][0] + 63) // 64) * ((group.layout[e[2]][1][1] + 63) // 64) for e in entries)
                n_splits = max(1, min((2048 + max(n_t, 1) - 1) // max(n_t, 1), max(rows // 256, 1), 32))
        arr = (L.DwEntryT * len(entries))()
        items: List[int] = []
        big_items: List[int] = []
        coop_items: List[int] = []
        work: List[int] = []
        s_full = max(1, min(rows // 512, 8)) if n_splits is None else int(n_splits)  # row splits of a full tile
        kinds: List[str] = []  # which launch takes entry i ("work" / "coop" / "big" / "items"): its row-split count
        for i, ent in enumerate(entries):
            dz, a, wk, bk = ent[:4]
            # optional 5th/6th items: (ptr, row stride, width) views for strided operands
            out_f, in_f = group.layout[wk][1]
            if len(ent) > 4:
                ldz, lda_ = ent[4], ent[5]
            else:
                ldz = lda_ = 0
                assert (dz.shape[1], a.shape[1]) == (out_f, in_f), (wk, group.layout[wk], dz.shape, a.shape)
            arr[i].ldz, arr[i].lda = ldz, lda_
            arr[i].dz = dz if isinstance(dz, int) else dz.data_ptr()
            arr[i].a = a if isinstance(a, int) else a.data_ptr()
            arr[i].w_off, arr[i].b_off = group.offset(wk), group.offset(bk)
            arr[i].out, arr[i].in_ = out_f, in_f
            if self.tile_blocks:
                T = self.tile_blocks
                ob_n, ib_n = (out_f + 15) // 16, (in_f + 15) // 16
                for ot in range((ob_n + T - 1) // T):
                    for it in range((ib_n + T - 1) // T):
                        blocks = min(T, ob_n - T * ot) * min(T, ib_n - T * it)
                        # a k-step costs max(load latency ~1.2 us, its MFMAs): a strip of few blocks is latency-bound,
                        # so its time is its k-step COUNT -- it takes the full tiles' split count, not a share of it
                        nsp = s_full
                        for sp in range(nsp):
                            work += [i, ot, it, sp | (nsp << 16)]
                kinds.append("work")
                continue
            # osrl_mlp_backward_dw_big fills its fragments with 16-byte loads: aligned operands, strides % 4 == 0
            aligned = (arr[i].dz % 16 == 0 and arr[i].a % 16 == 0 and (ldz or out_f) % 4 == 0 and (lda_ or in_f) % 4 == 0
                       and arr[i].w_off % 4 == 0)
            if use_big and self.COOP and out_f % 256 == 0 and in_f % 256 == 0 and aligned and rows % 16 == 0:
                # token-matrix sized GEMMs (CDT projections): every 256x256 tile to osrl_mlp_backward_dw_coop
                for ot in range(out_f // 256):
                    for it in range(in_f // 256):
                        coop_items += [i, ot, it, 0]
                kinds.append("coop")
                continue
            if use_big and out_f % 128 == 0 and in_f % 64 == 0 and aligned:
                # (128x64 tiles, one wave each: what the 256x256 form does not take) osrl_mlp_backward_dw_big
                for ot in range(out_f // 128):
                    for it in range(in_f // 64):
                        big_items += [i, ot, it, 0]
                kinds.append("big")
                continue
            kinds.append("items")
            for ot in range((out_f + 63) // 64):
                for it in range((in_f + 63) // 64):
                    items += [i, ot, it, 0]
        self._kinds = kinds
        self._ranges = [(int(arr[i].w_off), int(arr[i].out) * int(arr[i].in_), int(arr[i].b_off), int(arr[i].out))
                        for i in range(len(entries))]
        self.n_items = len(items) // 4
        self.n_big = len(big_items) // 4
        self.n_coop = len(coop_items) // 4
        self.n_work = len(work) // 4
        self.d_work = torch.tensor(work, dtype=torch.int32, device=device) if work else None
        # arrival counters of the fused dW + optimizer launch (launch_adam): one per tile, shared by its row splits
        self.d_tile_ids = self.d_counters = None
        if work:
            ids, seen = [], {}
            for j in range(self.n_work):
                key = tuple(work[4 * j:4 * j + 3])
                ids.append(seen.setdefault(key, len(seen)))
            self.d_tile_ids = torch.tensor(ids, dtype=torch.int32, device=device)
            self.d_counters = torch.zeros(max(len(seen), 1), dtype=torch.int32, device=device)
        self._adam_c = None
        raw = bytes(arr)
        self.d_entries = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.d_items = torch.tensor(items if items else [0, 0, 0, 0], dtype=torch.int32, device=device)
        self.d_big = torch.tensor(big_items, dtype=torch.int32, device=device) if big_items else None
        self.d_coop = torch.tensor(coop_items, dtype=torch.int32, device=device) if coop_items else None
        self._keep = [e[0] for e in entries] + [e[1] for e in entries] + [e[6:] for e in entries if len(e) > 6]
        if n_splits is None:
            # one workgroup per (tile, split): aim at >= 4 rounds of the 512 resident workgroups (2 per CU at
            # 66 KB LDS) so the tail round is small, but keep >= 64 rows (4 k-steps) per wave; each split costs
            # one slab write here and one slab read in the Adam kernel
            n_splits = max(1, min((2048 + max(self.n_items, 1) - 1) // max(self.n_items, 1), max(rows // 256, 1), 32))
        self.n_splits_small = n_splits if self.n_items else 0
        # big tiles: 4 waves (tiles) per workgroup, one workgroup per CU -> as many row splits as fill the 256 CUs once
        self.n_splits_big = 0
        if self.n_big:
            wgs = (self.n_big + 3) // 4
            best, best_eff = 1, 0.0
            for S in range(1, min(32, max(rows // 512, 1)) + 1):  # fill whole rounds of the 256 CUs
                eff = wgs * S / (256.0 * ((wgs * S + 255) // 256))
                if eff > best_eff + 1e-9:
                    best, best_eff = S, eff
            self.n_splits_big = best
        # 256x256 tiles: one 8-wave workgroup per CU -> the FEWEST row splits that fill whole rounds of the 256 CUs best
        self.n_splits_coop = 0
        if self.n_coop:
            best, best_eff = 1, 0.0
            for S in range(1, min(32, max(rows // 2048, 1)) + 1):
                eff = self.n_coop * S / (256.0 * ((self.n_coop * S + 255) // 256))
                if eff > best_eff + 1e-9:
                    best, best_eff = S, eff
            self.n_splits_coop = best
        self.n_splits = max(self.n_splits_small, self.n_splits_big, self.n_splits_coop, 1)
        if self.n_work:
            self.n_splits = max(w >> 16 for w in work[3::4])
        group.ensure_slabs(self.n_splits)

    def split_ranges(self) -> List[Tuple[int, int, int]]:
        """(float offset, length, row splits) of every parameter range this plan writes: the slabs a range's consumer has
        to sum (``launch()``: slabs beyond a range's own split count stay zero)."""
        per = {"work": self.n_splits, "coop": self.n_splits_coop, "big": self.n_splits_big, "items": self.n_splits_small}
        out = []
        for kind, (w_off, w_len, b_off, b_len) in zip(self._kinds, self._ranges):
            out += [(w_off, w_len, max(per[kind], 1)), (b_off, b_len, max(per[kind], 1))]
        return out

    def can_fuse_adam(self) -> bool:
        """The fused dW + optimizer launch covers a group whose whole plan is the flat (tile, split) work list."""
        flat = bool(self.n_work) and not self.n_items and not self.n_big
        if FUSE_DW_ADAM == "auto":
            return flat and self.n_splits == 1
        return flat and FUSE_DW_ADAM == "1"

    def launch_adam(self, lr: float, st_ptr: int, tau: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                    polyak: bool = True) -> None:
        """dW and the group's Adam (+ Polyak + packed-copy refresh) step in ONE launch
        (osrl_mlp_backward_dw_tiles_adam): same bits as ``launch()`` + ``group.adam_step(lr, st_ptr, tau)``.
        Single-device steps only -- a data-parallel step all-reduces the gradient in between."""
        g = self.group
        g.cur_splits = self.n_splits
        key = (float(lr), int(st_ptr), float(tau), tuple(betas), float(eps), bool(polyak))
        if self._adam_c is None or self._adam_c[0] != key:
            o = L.DwAdamT()
            o.p, o.m, o.v = g.p.data_ptr(), g.m.data_ptr(), g.v.data_ptr()
            o.tgt = _ptr(g.tgt) if (polyak and g.tgt is not None) else None
            if g.weights:
                o.map_f, o.map_b = g._map_f.data_ptr(), g._map_b.data_ptr()
                o.pf, o.pb = g.pf.data_ptr(), g.pb.data_ptr()
                o.tf = _ptr(g.tf) if o.tgt else None
            o.st = st_ptr
            o.lr, o.beta1, o.beta2, o.eps, o.tau = lr, betas[0], betas[1], eps, tau
            self._adam_c = (key, o)
        o = self._adam_c[1]
        L.check(L.load().osrl_mlp_backward_dw_tiles_adam(
            self.d_entries.data_ptr(), self.d_work.data_ptr(), self.d_tile_ids.data_ptr(), self.d_counters.data_ptr(),
            self.n_work, self.rows, self.tile_blocks, g.slabs.data_ptr(), g.n, C.byref(o), cur_stream()),
            "osrl_mlp_backward_dw_tiles_adam")

    def launch(self) -> None:
        """Both launches write disjoint parameter ranges; a range's slabs beyond its own split count are never written
        by the plans of ONE engine and are zero (FlatGroup.ensure_slabs allocates zeros and re-zeroes a retained tensor
        whenever a plan is attached), so the consumer may sum ``n_splits`` slabs of everything."""
        g = self.group
        g.cur_splits = self.n_splits
        if self.n_work:
            L.check(L.load().osrl_mlp_backward_dw_tiles(self.d_entries.data_ptr(), self.d_work.data_ptr(), self.n_work,
                                                        self.rows, self.tile_blocks, g.slabs.data_ptr(), g.n,
                                                        cur_stream()), "osrl_mlp_backward_dw_tiles")
            return
        if self.n_items:
            L.check(L.load().osrl_mlp_backward_dw(self.d_entries.data_ptr(), self.d_items.data_ptr(), self.n_items,
                                                  self.rows, self.n_splits_small, g.slabs.data_ptr(), g.n, cur_stream()),
                    "osrl_mlp_backward_dw")
        if self.n_coop:
            L.check(L.load().osrl_mlp_backward_dw_coop(self.d_entries.data_ptr(), self.d_coop.data_ptr(), self.n_coop,
                                                       self.rows, self.n_splits_coop, g.slabs.data_ptr(), g.n,
                                                       cur_stream()), "osrl_mlp_backward_dw_coop")
        if self.n_big:
            L.check(L.load().osrl_mlp_backward_dw_big(self.d_entries.data_ptr(), self.d_big.data_ptr(), self.n_big,
                                                      self.rows, self.n_splits_big, g.slabs.data_ptr(), g.n,
                                                      cur_stream()), "osrl_mlp_backward_dw_big")


class Branches:
    """Fork/join of independent parts of a step onto side streams.  Inside hipGraph capture this turns
    into parallel graph branches (the 2048-row kernels occupy <= 256 workgroups each, so independent
    phases overlap on the 256 CUs); disabled => everything runs in order on the current stream."""

    def __init__(self, enabled: bool, n: int = 2):
        self.enabled = enabled and torch.cuda.is_available()
        self.side = [torch.cuda.Stream() for _ in range(n)] if self.enabled else []

    def fork(self, i: int, after: Optional[int] = None) -> None:
        """side[i] starts after everything enqueued so far on the current stream (or on side[after])."""
        if self.enabled:
            self.side[i].wait_stream(torch.cuda.current_stream() if after is None else self.side[after])

    def on(self, i: int):
        import contextlib
        return torch.cuda.stream(self.side[i]) if self.enabled else contextlib.nullcontext()

    def join(self, i: int) -> None:
        if self.enabled:
            torch.cuda.current_stream().wait_stream(self.side[i])

    def mark(self, i: int):
        """Event after the work enqueued so far on side[i]."""
        if not self.enabled:
            return None
        ev = torch.cuda.Event()
        ev.record(self.side[i])
        return ev

    def wait(self, ev) -> None:
        if self.enabled and ev is not None:
            torch.cuda.current_stream().wait_event(ev)


class ArgArena:
    """Device-resident argument blocks for the fused-MLP launches of a captured step (csrc/argmem.h, include/osrl_amd.h
    ``osrl_args_begin``).  Usage around a graph capture::

        arena = ArgArena(device)
        with arena.record():      # the warm-up pass: descriptors are copied into the host staging buffer
            body()
        arena.upload()            # one host -> device copy, outside the capture
        with graph_capture(g), arena.replay():
            body()                # launches whose descriptor is in the arena read it from HBM
        keep = arena              # the device copy must outlive the graph

    Why: where the HIP runtime keeps kernel arguments in HOST memory (no large BAR, HIP_FORCE_DEV_KERNARG=0) every
    wave fetches its 1-2 KB descriptor over PCIe -- the CPQ step measured 1690 instead of 2150 steps/s.
    ``OSRL_ARG_ARENA=0`` disables it (A/B measurements)."""

    ENABLED = _plan.knob("OSRL_ARG_ARENA", "1", "launch descriptors of a captured step in device memory", operator=True) == "1"

    def __init__(self, device, capacity: int = 1 << 18):
        self.device = torch.device(device)
        self.host = torch.zeros(capacity, dtype=torch.uint8)
        self.dev: Optional[torch.Tensor] = None
        self.used = 0
        self.blocks = self.hits = self.misses = 0

    class _Ctx:
        def __init__(self, arena: "ArgArena", mode: int):
            self.a, self.mode, self.open = arena, mode, False

        def __enter__(self):
            a = self.a
            if not ArgArena.ENABLED:
                return a
            dev = None if a.dev is None else a.dev.data_ptr()
            L.check(L.load().osrl_args_begin(a.host.data_ptr(), dev, a.host.numel(), a.used, self.mode),
                    "osrl_args_begin")
            self.open = True
            return a

        def __exit__(self, *exc):
            if self.open:
                used, nb, nh, nm = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
                L.check(L.load().osrl_args_end(C.byref(used), C.byref(nb), C.byref(nh), C.byref(nm)), "osrl_args_end")
                a = self.a
                a.used = int(used.value)
                if self.mode == 1:
                    a.blocks += int(nb.value)
                else:
                    a.hits, a.misses = int(nh.value), int(nm.value)
                    if a.misses:  # a launch whose descriptor the record pass did not store runs by value: correct, but its
                        import warnings  # arguments then sit wherever the runtime keeps kernargs
                        warnings.warn(f"osrl_amd: {a.misses} launch descriptor(s) of a captured step were not found in the "
                                      f"argument arena ({a.hits} were); those launches take their arguments by value")
            return False

    def record(self) -> "ArgArena._Ctx":
        return ArgArena._Ctx(self, 1)

    def upload(self) -> None:
        if ArgArena.ENABLED and self.used:
            self.dev = self.host[:self.used].to(self.device)  # synchronous copy from pageable memory

    def replay(self) -> "ArgArena._Ctx":
        if ArgArena.ENABLED and self.dev is None:
            self.dev = torch.zeros(64, dtype=torch.uint8, device=self.device)  # nothing recorded: every lookup misses
        return ArgArena._Ctx(self, 2)


def graph_capture(g: "torch.cuda.CUDAGraph"):
    """``torch.cuda.graph(g)`` in THREAD-LOCAL capture mode.  The default (global) mode makes a capture fail when ANY thread
    of the process calls a capture-unsafe runtime function meanwhile -- and a process that holds an NCCL (= RCCL) process
    group has such a thread: the group's watchdog polls the events of collectives still in flight (``hipEventQuery``), e.g.
    those of the warm-up pass a data-parallel engine runs right before it captures its step.  Seen once in four runs of the
    GPU suite: "operation not permitted when stream is capturing" thrown on the watchdog thread, which aborts the process.
    Thread-local mode restricts the check to the capturing thread, which is the one that must not make such calls.

    ... and with Python's cyclic garbage collector OFF for the length of the capture (round 6, third session).  ``torch.cuda.graph``
    collects once when it is entered; a step body then creates enough objects (events, ctypes descriptors) to start an AUTOMATIC
    collection in the middle of the capture, and when that collection finds an engine of an earlier owner -- engines, their
    pipelines and their CUDAGraphs hold each other in cycles -- it destroys a hipGraph / its streams on the capturing thread:
    "operation not permitted when stream is capturing" inside a destructor, i.e. ``abort()``.  Deterministic for a given sequence
    of allocations: one pytest command line aborted 5 times of 5 in its 25th test ("Fatal Python error: Aborted ...
    Garbage-collecting" under ``prologue`` inside ``PipelinedSteps._capture_once``) while the same tests under ``-v`` and the
    full suite never did (DESIGN_LOG)."""
    return _GraphCapture(g)


class _GraphCapture:
    def __init__(self, g):
        self.g, self.ctx, self.gc_was = g, None, False

    def __enter__(self):
        import gc
        self.gc_was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            self.ctx = torch.cuda.graph(self.g, capture_error_mode="thread_local")
            return self.ctx.__enter__()
        except BaseException:
            if self.gc_was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.gc_was:
                gc.enable()


# A two-branch hipGraph runs its side branch on a stream the runtime creates when the graph is instantiated, and the
# runtime maps streams onto its few hardware queues in creation order: depending on how many streams the process made
# before, the two branches land on two queues (they overlap) or on ONE (they serialise).  Measured in round 6
# (DESIGN_LOG): the same one-step CPQ graph captured after a PipelinedSteps had been built on its engine replayed at
# 2020-2050 steps/s through the Trainer API against 2230-2265 on a fresh engine -- identical host profile, identical
# kernels; GPU_MAX_HW_QUEUES=8 instead of 4 moved the fresh engine to 1293.  So a capture is repeated a few times (every
# repetition creates three streams, i.e. shifts the mapping) and the FASTEST graph is kept.
# Measured (gpurun_out/r6r): best-of-4 does NOT recover that case (2018 vs 1996) and changes nothing elsewhere (C2 2318-2326
# vs 2303-2328, C3 / C4 equal) -- whatever slows that graph is not the draw of the mapping.  Default 1 (= one capture); the
# helper stays as a lab switch.
CAPTURE_TRIES = int(_plan.knob("OSRL_CAPTURE_TRIES", "1", "captures of a two-branch step graph to pick the fastest from"))


def pick_fastest(build, replay, snapshot, restore, tries: int, reps: int = 12):
    """``build()`` -> a captured candidate (any object); ``replay(c)`` enqueues one replay of it.  Returns the candidate
    whose ``reps`` replays took least (HIP events; the training state the timing replays advance is put back), and the
    list of all times in ms."""
    if tries <= 1:
        return build(), []
    cands, times = [], []
    for _ in range(tries):
        c = build()
        snap = snapshot()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            replay(c)
        e0.record()
        for _ in range(reps):
            replay(c)
        e1.record()
        torch.cuda.synchronize()
        restore(snap)
        cands.append(c)
        times.append(e0.elapsed_time(e1) / reps)
    return cands[min(range(tries), key=lambda i: times[i])], times


def capture_step(device, warm, captured):
    """hipGraph capture of one step: ``warm()`` runs eagerly on a side stream (torch's warm-up requirement) while
    the fused-MLP launches' argument blocks are recorded, the blocks go to HBM, then ``captured()`` is captured with
    those launches reading their descriptors from there (ArgArena).  Returns (graph, arena); keep both alive."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    arena = ArgArena(device)
    with torch.cuda.stream(s), arena.record():
        warm()
    torch.cuda.current_stream().wait_stream(s)
    arena.upload()
    g = torch.cuda.CUDAGraph()
    with graph_capture(g), arena.replay():
        captured()
    return g, arena


FUSED_BEGIN = _plan.knob("OSRL_FUSED_BEGIN", "1", "tick + minibatch gather + noise as one prologue launch") == "1"


class StepState:
    """Device-resident step counter / bias corrections / statistics ring (csrc/optim.hip)."""

    def __init__(self, device, stat_keys: Sequence[str], ring_len: int = 256, betas=(0.9, 0.999), warmup: int = 0):
        self.keys = list(stat_keys)
        self.index = {k: i for i, k in enumerate(self.keys)}
        self.n_stats = max(len(self.keys), 1)
        self.ring_len = ring_len
        self.betas, self.warmup = betas, warmup
        self.state = torch.zeros(24, dtype=torch.uint8, device=device)
        self.stats = torch.zeros(self.n_stats, dtype=torch.float32, device=device)
        self.ring = torch.zeros(ring_len, self.n_stats, dtype=torch.float32, device=device)
        self._clock = [0]  # host mirror of the train-step count; shared with a linked peer (link())
        self.peer: Optional["StepState"] = None
        # callables run at every point where the host synchronises with the step anyway (statistics reads, device_step):
        # an engine whose launches can flag a failure on device (the one-launch BC step's bounded wait) registers its
        # check here, so the flag cannot go unread for longer than one statistics flush
        self.health_checks: list = []

    @property
    def host_step(self) -> int:
        return self._clock[0]

    @host_step.setter
    def host_step(self, v: int) -> None:
        self._clock[0] = int(v)

    def link(self, other: "StepState") -> None:
        """Make ``other`` this state's PEER: the two take turns (software-pipelined steps, engine/pipeline.py -- step k+1's
        prologue ticks one state while step k's last optimizer launches still read the other's bias corrections).  Every
        tick of either gives it ``max(own, peer) + 1`` (osrl_step_tick_peer / osrl_step_begin_peer), so any interleaving of
        the two -- strict alternation inside a pipelined graph, single steps on one of them in between -- counts 1, 2, 3,
        ...  They share the host step mirror and ONE statistics ring (slot = step - 1: no collisions); each keeps its own
        ``stats`` buffer, committed to the ring at its own next tick."""
        assert other is not self and other.n_stats == self.n_stats and other.ring_len == self.ring_len
        assert (other.betas, other.warmup) == (self.betas, self.warmup)
        self.peer, other.peer = other, self
        other._clock = self._clock
        other.ring = self.ring
        other.state.copy_(self.state)
        other.state[20:24].zero_()  # (its own arrival counter)

    def _peer_ptr(self) -> Optional[int]:
        return None if self.peer is None else self.peer.state.data_ptr()

    def _health(self) -> None:
        for f in self.health_checks:
            f()
        if self.peer is not None:
            for f in self.peer.health_checks:
                f()

    @property
    def ptr(self) -> int:
        return self.state.data_ptr()

    def stat_ptr(self, key: str) -> int:
        return self.stats.data_ptr() + 4 * self.index[key]

    def tick(self) -> None:
        L.check(L.load().osrl_step_tick_peer(self.ptr, self._peer_ptr(), self.betas[0], self.betas[1], self.warmup,
                                             self.stats.data_ptr(), self.ring.data_ptr(), self.n_stats, self.ring_len,
                                             cur_stream()), "osrl_step_tick")
        self.host_step += 1

    def begin(self, noise: Optional[torch.Tensor] = None, noise_seed: int = 0, noise_stream: int = 0,
              gather=None) -> None:
        """The step prologue in ONE launch (csrc/rng.hip ``step_begin_kernel``): this tick + the Philox fill of
        ``noise`` (flat fp32 buffer, the draws of ``randn_fill(noise, noise_seed, noise_stream, st)``) + the minibatch
        gather described by ``gather`` = ``store.gather_args(dst)``.  Same results as the three separate launches."""
        n_f, src, dst, w, sc, n_rows, B, g_seed, g_stream, _keep = gather if gather is not None else \
            (0, None, None, None, None, 0, 0, 0, 0, None)
        L.check(L.load().osrl_step_begin_peer(
            self.ptr, self._peer_ptr(), self.betas[0], self.betas[1], self.warmup, self.stats.data_ptr(), self.ring.data_ptr(),
            self.n_stats, self.ring_len, None if noise is None else noise.data_ptr(),
            0 if noise is None else noise.numel(), noise_seed, noise_stream, n_f, src, dst, w, sc, n_rows, B, g_seed,
            g_stream, cur_stream()), "osrl_step_begin")
        self.host_step += 1

    def prologue(self, replay, dst, noise: Optional[torch.Tensor], seed: int, device_noise: bool,
                 fields: Optional[Sequence[int]] = None) -> None:
        """tick [+ minibatch gather from ``replay`` into ``dst``] [+ noise fill]: one fused launch when there is
        anything beside the tick (OSRL_FUSED_BEGIN=0: the separate launches, for A/B measurements)."""
        want_noise = device_noise and noise is not None
        if FUSED_BEGIN and (replay is not None or want_noise):
            self.begin(noise if want_noise else None, seed, 0,
                       None if replay is None else replay.gather_args(dst, fields))
            return
        self.tick()
        if replay is not None:
            if fields is None:
                replay.gather(dst, self.ptr)
            else:
                replay.gather_fields(fields, dst, self.ptr)
        if want_noise:
            randn_fill(noise, seed, 0, self.ptr)

    def set_step(self, step: int) -> None:
        """Continue counting from ``step`` completed train steps (engine rebuild, checkpoint resume): the next
        tick makes it step+1 and recomputes the bias corrections / warm-up scale from that."""
        self.state[:8].view(torch.int64).fill_(int(step))
        if self.peer is not None:  # (max(own, peer) + 1 is then step + 1 whichever of the two ticks next)
            self.peer.state[:8].view(torch.int64).fill_(int(step))
        self.host_step = int(step)

    def _own_step(self) -> int:
        return int(self.state[:8].view(torch.int64).item())

    def device_step(self) -> int:
        v = self._own_step()
        if self.peer is not None:
            v = max(v, self.peer._own_step())
        self._health()
        return v

    def _holder(self, step: int) -> Optional["StepState"]:
        """The state whose ``stats`` buffer holds train step ``step`` (not yet committed to the ring), or None."""
        if self.peer is None:
            return self if step == self.host_step else None
        for c in (self, self.peer):
            if c._own_step() == step:
                return c
        return None

    def read_stats(self, step: Optional[int] = None) -> Dict[str, float]:
        """Statistics of train step ``step`` (1-based; default = the latest).  Synchronises."""
        self._health()
        if step is None:
            step = self.host_step
        h = self._holder(step)
        if h is not None:
            v = h.stats.tolist()
        else:
            if self.host_step - step >= self.ring_len:
                raise RuntimeError("statistics of that step were already overwritten in the ring")
            v = self.ring[(step - 1) % self.ring_len].tolist()
        return dict(zip(self.keys, v))

    def read_stats_many(self, steps) -> Dict[int, List[float]]:
        """Statistics rows of several past steps through ONE device->host copy of the ring (the lazy logger's flush:
        per-row reads cost one synchronising copy each, ~95 us per train step amortised at 5 keys)."""
        steps = list(steps)
        if not steps:
            return {}
        if min(steps) < 1 or self.host_step - min(steps) >= self.ring_len:
            raise RuntimeError("statistics of that step were already overwritten in the ring")
        if self.peer is not None:  # two uncommitted steps may be outstanding: which state holds which is read off the device
            held = {c._own_step(): c for c in (self.peer, self)}
            both = torch.cat([self.ring.reshape(-1), self.stats, self.peer.stats]).tolist()
            self._health()
            n = self.n_stats
            bufs = {id(self): both[self.ring_len * n:(self.ring_len + 1) * n], id(self.peer): both[(self.ring_len + 1) * n:]}
            out = {}
            for s_ in steps:
                if s_ in held:
                    out[s_] = bufs[id(held[s_])]
                else:
                    r = (s_ - 1) % self.ring_len
                    out[s_] = both[r * n:(r + 1) * n]
            return out
        both = torch.cat([self.ring.reshape(-1), self.stats]).tolist()  # one kernel, one synchronising copy
        self._health()
        n = self.n_stats
        cur = both[self.ring_len * n:]
        out = {}
        for s_ in steps:
            if s_ == self.host_step:
                out[s_] = cur
            else:
                r = (s_ - 1) % self.ring_len
                out[s_] = both[r * n:(r + 1) * n]
        return out


def randn_fill(out: torch.Tensor, seed: int, stream_id: int, st_ptr: Optional[int]) -> None:
    L.check(L.load().osrl_randn_fill(out.data_ptr(), out.numel(), seed, stream_id, st_ptr, cur_stream()),
            "osrl_randn_fill")


def load_into(pairs) -> None:
    """Copy caller tensors into the engine's static batch buffers (the addresses the captured graph reads).  Device
    fp32 sources of the right size go through ONE multi-tensor copy launch instead of one launch per tensor -- this is
    the per-call cost ``trainer.train_one_step(tensors)`` adds on top of the replayed graph; anything else (host
    arrays, other dtypes) takes the per-tensor ``copy_``."""
    fast_d, fast_s = [], []
    for dst, src in pairs:
        if src is dst:
            continue
        if isinstance(src, torch.Tensor) and src.device == dst.device and src.dtype == dst.dtype \
                and src.numel() == dst.numel() and src.is_contiguous():
            fast_d.append(dst)
            fast_s.append(src.view(dst.shape))
        else:
            dst.copy_(torch.as_tensor(src).reshape(dst.shape), non_blocking=True)
    if len(fast_d) == 1:
        fast_d[0].copy_(fast_s[0], non_blocking=True)
    elif fast_d:
        torch._foreach_copy_(fast_d, fast_s, non_blocking=True)
