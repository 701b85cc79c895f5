"""GPU unit parity of the individual HIP kernels (through the C ABI via ctypes) against fp64
numpy/torch-CPU references of the same op.  Tolerances: fp32 round-off (1e-5 relative-ish)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _mk_nets(rs, E, dims, dev):
    nets = []
    for _ in range(E):
        layers = []
        for l in range(len(dims) - 1):
            k = 1 / math.sqrt(dims[l])
            W = torch.tensor(rs.uniform(-k, k, (dims[l + 1], dims[l])), dtype=torch.float32, device=dev)
            b = torch.tensor(rs.uniform(-k, k, (dims[l + 1],)), dtype=torch.float32, device=dev)
            layers.append((W, b))
        nets.append(layers)
    return nets


def _act64(name, x):
    return {"relu": lambda v: np.maximum(v, 0), "tanh": np.tanh, "id": lambda v: v}[name](x)


def _dact64(name, y):
    return {"relu": lambda v: (v > 0).astype(np.float64), "tanh": lambda v: 1 - v * v,
            "id": lambda v: np.ones_like(v)}[name](y)


MLP_CASES = [
    # E, dims, acts, out_scale, rows, (d0, map0, div0, map1, div1), dx_cols
    (2, [78, 256, 256, 1], ["relu", "relu", "id"], 1.0, 100, (76, 0, 1, 0, 1), (76, 2)),
    (1, [80, 400, 400, 2], ["relu", "relu", "tanh"], 1.5, 77, (76, 0, 1, 0, 1), (76, 4)),
    (1, [78, 400, 400, 8], ["relu", "relu", "id"], 1.0, 2048, (76, 0, 1, 0, 1), None),
    (3, [9, 32, 24, 3], ["tanh", "tanh", "tanh"], 1.0, 1, (6, 0, 1, 0, 1), (0, 9)),
    (4, [41, 256, 256, 1], ["relu", "relu", "id"], 1.0, 333, (33, 2, 3, 0, 1), (33, 8)),     # DIV map
    (2, [78, 256, 256, 1], ["relu", "relu", "id"], 1.0, 4096 + 40, (76, 1, 517, 0, 1), None),  # MOD map, NRB=4
    (1, [76, 256, 256, 4], ["relu", "relu", "id"], 1.0, 2048, (76, 0, 1, 0, 1), None),       # no second source
    (8, [12, 64, 64, 1], ["relu", "relu", "id"], 1.0, 50, (8, 0, 1, 0, 1), (8, 4)),
    (1, [7, 48, 5], ["relu", "id"], 1.0, 19, (7, 0, 1, 0, 1), None),                          # 2 layers
    (1, [20, 64, 48, 32, 6], ["relu", "tanh", "relu", "tanh"], 2.0, 130, (16, 0, 1, 0, 1), (3, 11)),  # 4 layers
]


def _random_mlp_cases(n, seed=2024):
    """Seeded random shapes inside the C ABI's limits (1-4 layers, widths 1..448, 1-8 nets, 1..6000 rows, every row
    map, with / without a second source and a dx slice): shapes no engine uses must work too -- the tile choice
    (16/32/64-row tiles, 4/8 waves, narrow heads, capped loop kernel) is a function of the shape."""
    rs = np.random.RandomState(seed)
    acts_all = ["relu", "tanh", "id"]
    out = []
    for i in range(n):
        L_ = int(rs.randint(1, 5))
        widths = [int(rs.choice([1, 3, 8, 17, 32, 64, 100, 256, 300, 400, 448])) for _ in range(L_)]
        k0 = int(rs.choice([1, 2, 5, 16, 33, 78, 130]))
        dims = [k0] + widths
        acts = [acts_all[int(rs.randint(0, 3))] for _ in range(L_)]
        E = int(rs.choice([1, 1, 2, 3, 4, 8]))
        rows = int(rs.choice([1, 2, 15, 16, 17, 31, 100, 513, 2048, 2500, 6000]))
        if E * rows * max(dims) > 6_000_000:  # keep the fp64 reference cheap
            rows = max(1, 6_000_000 // (E * max(dims)))
        d0 = int(rs.randint(1, k0 + 1))
        mode = int(rs.randint(0, 3))
        div0 = {0: 1, 1: int(rs.randint(1, rows + 1)), 2: int(rs.randint(1, 12))}[mode]
        dxc = None
        if rs.rand() < 0.6:
            c0 = int(rs.randint(0, k0))
            dxc = (c0, int(rs.randint(1, k0 - c0 + 1)))
        out.append((E, dims, acts, float(rs.choice([1.0, 0.5, 2.0])), rows, (d0, mode, div0, 0, 1), dxc))
    return out


MLP_CASES += _random_mlp_cases(28)


@pytest.mark.parametrize("ci", range(len(MLP_CASES)))
def test_mlp_fwd_bwd(ci):
    from osrl_amd.engine.core import DwPlan, FlatGroup, LayerRef, MlpRun, NetDesc
    E, dims, acts, oscale, rows, (d0, map0, div0, map1, div1), dxc = MLP_CASES[ci]
    dev = _dev()
    rs = np.random.RandomState(10 + ci)
    # parameters in a FlatGroup so dW offsets are exercised
    grp = FlatGroup("t", dev)
    for e in range(E):
        for l in range(len(dims) - 1):
            grp.add(f"{e}.{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{e}.{l}.w")
            grp.add(f"{e}.{l}.b", (dims[l + 1],))
    grp.finalize()
    nets, refs = [], []
    for e in range(E):
        layers, rr = [], []
        for l in range(len(dims) - 1):
            k = 1 / math.sqrt(dims[l])
            W, b = grp.view(f"{e}.{l}.w"), grp.view(f"{e}.{l}.b")
            W.copy_(torch.tensor(rs.uniform(-k, k, W.shape), dtype=torch.float32))
            b.copy_(torch.tensor(rs.uniform(-k, k, b.shape), dtype=torch.float32))
            layers.append((W, b))
            rr.append(LayerRef(W, b, grp, f"{e}.{l}.w", f"{e}.{l}.b"))
        nets.append(layers)
        refs.append(rr)
    grp.repack()
    desc = NetDesc(refs, acts, oscale)
    d1 = dims[0] - d0
    n0 = {0: rows, 1: div0, 2: (rows + div0 - 1) // div0}[map0]
    src0 = torch.tensor(rs.randn(n0, d0), dtype=torch.float32, device=dev)
    src1 = torch.tensor(rs.randn(rows, d1), dtype=torch.float32, device=dev) if d1 else None
    run = MlpRun(desc, rows, True, dev)
    y = run.forward(src0, src1, map0=map0, div0=div0, map1=map1, div1=div1)
    torch.cuda.synchronize()

    # fp64 reference
    idx0 = {0: np.arange(rows), 1: np.arange(rows) % div0, 2: np.arange(rows) // div0}[map0]
    X = src0.cpu().numpy().astype(np.float64)[idx0]
    if d1:
        X = np.concatenate([X, src1.cpu().numpy().astype(np.float64)], 1)
    np.testing.assert_allclose(run.x.cpu().numpy(), X, atol=0, rtol=0, err_msg="saved input x")
    caches = []
    for e in range(E):
        h, c = X, [X]
        for l, a in enumerate(acts):
            W, b = (t.cpu().numpy().astype(np.float64) for t in nets[e][l])
            h = _act64(a, h @ W.T + b)
            if l == len(acts) - 1:
                h = h * oscale
            got = run.h[e][l].cpu().numpy()
            err = np.abs(got - h).max()
            assert err < 2e-5 * max(1.0, np.abs(h).max()), f"case {ci} fwd net {e} layer {l}: max err {err}"
            # continue (and later differentiate) from the GPU's own activations so that ReLU masks of
            # pre-activations within fp32 round-off of 0 cannot flip between the two computations
            h = got.astype(np.float64)
            c.append(h)
        caches.append(c)
    assert np.isfinite(y.cpu().numpy()).all()

    # backward
    dy = torch.tensor(rs.randn(E, rows, dims[-1]), dtype=torch.float32, device=dev)
    run.setup_backward(dy, need_dz=True, dx_cols=dxc)
    run.backward_dz()
    plan = DwPlan(grp, run.dw_entries(), rows, dev)
    plan.launch()
    torch.cuda.synchronize()
    for e in range(E):
        c = caches[e]
        g = dy[e].cpu().numpy().astype(np.float64)
        L_ = len(acts)
        for l in range(L_ - 1, -1, -1):
            yl = c[l + 1] / (oscale if l == L_ - 1 else 1.0)
            dz = g * _dact64(acts[l], yl) * (oscale if l == L_ - 1 else 1.0)
            got = run.dz[e][l].cpu().numpy()
            sc = max(1.0, np.abs(dz).max())
            assert np.abs(got - dz).max() < 3e-5 * sc, f"case {ci} dz net {e} layer {l}: {np.abs(got - dz).max()}"
            W = nets[e][l][0].cpu().numpy().astype(np.float64)
            dW, db = dz.T @ c[l], dz.sum(0)
            gW = grp.grad_view(f"{e}.{l}.w").cpu().numpy()
            gb = grp.grad_view(f"{e}.{l}.b").cpu().numpy()
            scw = max(1.0, np.abs(dW).max())
            assert np.abs(gW - dW).max() < 1e-4 * scw, f"case {ci} dW net {e} layer {l}: {np.abs(gW - dW).max()} / {scw}"
            assert np.abs(gb - db).max() < 1e-4 * max(1.0, np.abs(db).max()), f"case {ci} db net {e} layer {l}"
            g = dz @ W
        if dxc is not None:
            c0, nc = dxc
            got = run.dx[e].cpu().numpy()
            ref = g[:, c0:c0 + nc]
            assert np.abs(got - ref).max() < 3e-5 * max(1.0, np.abs(ref).max()), f"case {ci} dx net {e}"
    # the same weight gradients through the flat (tile, split) work list on 80 x 80 and 64 x 64 tiles
    # (osrl_mlp_backward_dw_tiles): tiles of unequal block counts get unequal row splits
    ref_w = {k: grp.grad_view(k).cpu().numpy().astype(np.float64) for k in grp.layout}
    for T in (5, 4):
        plan_t = DwPlan(grp, run.dw_entries(), rows, dev, tile_blocks=T)
        assert plan_t.n_work > 0
        plan_t.launch()
        torch.cuda.synchronize()
        for k, ref in ref_w.items():
            got = grp.grad_view(k).cpu().numpy()
            assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), f"case {ci} tiles {T}: {k}"


BIG_CASES = [
    # E, dims, acts, out_scale, rows, (d0, map0, div0)          -- forward-only launches with >= 4 row blocks per CU
    (2, [78, 256, 256, 1], ["relu", "relu", "id"], 1.0, 20480, (76, 1, 2048)),     # C2 target cost-critics (80-row tiles)
    (1, [78, 400, 400, 8], ["relu", "relu", "id"], 1.0, 20480, (76, 1, 2048)),     # C2 VAE encoder (48/32-row tiles)
    (4, [41, 256, 256, 1], ["relu", "relu", "id"], 1.0, 8192 + 37, (33, 2, 10)),   # DIV map, ragged rows, 4 nets
    (1, [80, 400, 400, 2], ["relu", "relu", "tanh"], 1.5, 16384 + 5, (76, 0, 1)),  # tanh * scale head, ragged
    (2, [24, 96, 80, 40], ["tanh", "relu", "id"], 1.0, 33000, (20, 0, 1)),          # non-narrow last layer, odd widths
    (1, [76, 256, 256, 24], ["relu", "relu", "id"], 1.0, 30000, (76, 0, 1)),        # 2-block narrow head, no 2nd source
    (8, [12, 64, 64, 1], ["relu", "relu", "id"], 1.0, 6000, (8, 0, 1)),             # 8 small nets
    # widths that are not multiples of 16 in the 80-row kernel: 13 / 16 column blocks (waves with 4 and with 3 blocks),
    # tanh in a wide layer, zero-padded k of the next layer
    (1, [78, 200, 250, 6], ["tanh", "relu", "id"], 1.0, 20480 + 3, (76, 1, 2048)),
    (1, [40, 420, 440, 3], ["relu", "tanh", "id"], 2.0, 10240, (33, 0, 1)),         # 27 / 28 blocks: the 7-block form
    # round 6, the ONE-output head fused into the last wide layer's epilogue (nb_head_dot): a single wide layer in front of
    # it; ragged widths (14 / 15 blocks: waves with 4 and 3 blocks) with tanh in the last wide layer, a tanh * scale head
    (2, [23, 256, 1], ["relu", "id"], 1.0, 9000, (17, 0, 1)),
    (3, [41, 250, 230, 1], ["tanh", "tanh", "tanh"], 2.0, 12000 + 11, (33, 2, 10)),
]


@pytest.mark.parametrize("ci", range(len(BIG_CASES)))
def test_mlp_fwd_big_rows(ci):
    """The 80-row one-workgroup-per-CU forward kernel (mlp_fwd_nb_kernel, tile_rows=80) vs fp64 and vs the tile kernels
    (tile_rows = 32 / 16); forward-only launches, as the N*B-row launches of CPQ / BCQ-Lag."""
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
    E, dims, acts, oscale, rows, (d0, map0, div0) = BIG_CASES[ci]
    dev = _dev()
    rs = np.random.RandomState(100 + ci)
    grp = FlatGroup("t", dev)
    for e in range(E):
        for l in range(len(dims) - 1):
            grp.add(f"{e}.{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{e}.{l}.w")
            grp.add(f"{e}.{l}.b", (dims[l + 1],))
    grp.finalize()
    refs, Ws = [], []
    for e in range(E):
        rr, ww = [], []
        for l in range(len(dims) - 1):
            k = 1 / math.sqrt(dims[l])
            W, b = grp.view(f"{e}.{l}.w"), grp.view(f"{e}.{l}.b")
            W.copy_(torch.tensor(rs.uniform(-k, k, W.shape), dtype=torch.float32))
            b.copy_(torch.tensor(rs.uniform(-k, k, b.shape), dtype=torch.float32))
            rr.append(LayerRef(W, b, grp, f"{e}.{l}.w", f"{e}.{l}.b"))
            ww.append((W.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)))
        refs.append(rr)
        Ws.append(ww)
    grp.repack()
    d1 = dims[0] - d0
    n0 = {0: rows, 1: div0, 2: (rows + div0 - 1) // div0}[map0]
    src0 = torch.tensor(rs.randn(n0, d0), dtype=torch.float32, device=dev)
    src1 = torch.tensor(rs.randn(rows, d1), dtype=torch.float32, device=dev) if d1 else None
    outs = []
    # (negative wave counts: the same form with OSRL_NB_HEAD=0 -- a one-output head run as a layer, as before round 6)
    for tile_rows, waves in ((32, 0), (16, 0), (80, 0), (80, 4), (80, 8), (80, 64), (80, -4), (80, -8), (80, -64)):  # 80: one workgroup per CU (shapes the kernel does
        # not take -- hidden layers of neither 13-16 nor 25-28 column blocks, wide last layer -- fall back to the default
        # tile); 25-block layers take its 8-wave form unless OSRL_NB_WAVES=4 asks for one wave per SIMD, 13-16-block
        # layers take theirs when OSRL_NB256_WAVES=8 asks for it and the 64-row form (two workgroups per CU) from 512
        # tiles per net on or when OSRL_NB64=1 asks for it
        desc = NetDesc(refs, acts, oscale)
        desc.c.tile_rows = tile_rows
        run = MlpRun(desc, rows, False, dev)
        if waves < 0:
            os.environ["OSRL_NB_HEAD"] = "0"
        if abs(waves) == 64:
            os.environ["OSRL_NB64"] = "1"
        elif waves:
            os.environ["OSRL_NB_WAVES"] = os.environ["OSRL_NB256_WAVES"] = str(abs(waves))
        try:
            y = run.forward(src0, src1, map0=map0, div0=div0)
        finally:
            os.environ.pop("OSRL_NB_WAVES", None)
            os.environ.pop("OSRL_NB256_WAVES", None)
            os.environ.pop("OSRL_NB64", None)
            os.environ.pop("OSRL_NB_HEAD", None)
        torch.cuda.synchronize()
        outs.append(torch.stack([t.clone() for t in y]).cpu().numpy() if isinstance(y, (list, tuple)) else y.clone().cpu().numpy())
    idx0 = {0: np.arange(rows), 1: np.arange(rows) % div0, 2: np.arange(rows) // div0}[map0]
    X = src0.cpu().numpy().astype(np.float64)[idx0]
    if d1:
        X = np.concatenate([X, src1.cpu().numpy().astype(np.float64)], 1)
    for e in range(E):
        h = X
        for l, a in enumerate(acts):
            h = _act64(a, h @ Ws[e][l][0].T + Ws[e][l][1])
        h = h * oscale
        for nm, got in (("tile32", outs[0][e]), ("tile16", outs[1][e]), ("tile80", outs[2][e]), ("tile80w4", outs[3][e]),
                        ("tile80w8", outs[4][e]), ("tile64", outs[5][e]), ("tile80w4 head as a layer", outs[6][e]),
                        ("tile80w8 head as a layer", outs[7][e]), ("tile64 head as a layer", outs[8][e])):
            err = np.abs(got.reshape(h.shape) - h).max()
            assert err < 3e-5 * max(1.0, np.abs(h).max()), f"case {ci} {nm} kernel net {e}: max err {err}"
    assert np.abs(outs[0] - outs[1]).max() < 3e-5 * max(1.0, np.abs(outs[1]).max())


SHARE_CASES = [
    # E, dims, acts, out_scale, src0 rows B, copies N, d0, form
    (2, [78, 256, 256, 1], ["relu", "relu", "id"], 1.0, 256, 10, 76, "80"),    # C2's target cost critics: 4 of 5 k-steps once per observation
    (1, [78, 400, 400, 16], ["relu", "relu", "id"], 1.0, 256, 10, 76, "80"),   # C2's VAE encoder: the 8-wave 25-block form
    (2, [23, 256, 256, 1], ["relu", "relu", "id"], 1.0, 304, 5, 17, "80"),     # C4: od = 17 -> one of two k-steps
    (2, [41, 250, 230, 1], ["tanh", "tanh", "tanh"], 2.0, 160, 15, 33, "80"),  # ragged widths, tanh, scaled head, 3 sample groups
    (3, [41, 256, 256, 4], ["relu", "relu", "id"], 1.0, 512, 8, 33, "64"),     # the 64-row form: 4 copies per tile
    (1, [78, 400, 400, 16], ["relu", "relu", "id"], 1.0, 250, 10, 76, "80"),   # B % 16 != 0: the hint is not taken, plain tiles run
]


@pytest.mark.parametrize("ci", range(len(SHARE_CASES)))
def test_mlp_fwd_big_rows_shared_src0_tiles(ci):
    """osrl_rows_t.share0: the rows of CPQ's N*B-row inference launches are the B observations N times over (row = n B + b,
    cpq.py:164-176), so a tile of [copies] x [16 observations] computes the part of layer 0 that lies inside the observation
    columns once per observation (csrc/mlp_nb.hip nb_share_acc) and walks only the remaining k-steps per row.  Same products,
    another order of a row's sum: within fp32 rounding of the plain launch (gate 2e-6 of the output scale; both forms within
    the fp64 gate of test_mlp_fwd_big_rows), incl. the KL rows the encoder launch leaves as its tail.  The switch is a hint:
    a shape that does not tile (case 5) runs the plain form -- bit-equal."""
    from osrl_amd.engine import glue as G
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
    E, dims, acts, oscale, B, N, d0, form = SHARE_CASES[ci]
    dev = _dev()
    rs = np.random.RandomState(300 + ci)
    grp = FlatGroup("t", dev)
    for e in range(E):
        for l in range(len(dims) - 1):
            grp.add(f"{e}.{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{e}.{l}.w")
            grp.add(f"{e}.{l}.b", (dims[l + 1],))
    grp.finalize()
    refs, Ws = [], []
    for e in range(E):
        rr, ww = [], []
        for l in range(len(dims) - 1):
            k = 1 / math.sqrt(dims[l])
            W, b = grp.view(f"{e}.{l}.w"), grp.view(f"{e}.{l}.b")
            W.copy_(torch.tensor(rs.uniform(-k, k, W.shape), dtype=torch.float32))
            b.copy_(torch.tensor(rs.uniform(-k, k, b.shape), dtype=torch.float32))
            rr.append(LayerRef(W, b, grp, f"{e}.{l}.w", f"{e}.{l}.b"))
            ww.append((W.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)))
        refs.append(rr)
        Ws.append(ww)
    grp.repack()
    rows = N * B
    src0 = torch.tensor(rs.randn(B, d0), dtype=torch.float32, device=dev)
    src1 = torch.tensor(rs.randn(rows, dims[0] - d0), dtype=torch.float32, device=dev)
    desc = NetDesc(refs, acts, oscale)
    desc.c.tile_rows = 80
    run = MlpRun(desc, rows, False, dev)
    k16 = d0 // 16
    kl_tail = dims[-1] == 16  # (the VAE encoder's launch: KL rows as its tail, osrl_mlp_forward_tail)
    kl_a, kl_b = torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
    os.environ["OSRL_NB64"] = "1" if form == "64" else "0"
    try:
        plain = torch.stack([t.clone() for t in run.forward(src0, src1, map0=1, div0=B,
                                                            tail=G.vae_kl_tail(8, kl_a) if kl_tail else None)]).cpu().numpy()
        assert (run.share_k16(d0, B, N) == k16) == (B % 16 == 0 and N % 5 == 0)
        got = torch.stack([t.clone() for t in run.forward(src0, src1, map0=1, div0=B, share_k16=k16,
                                                          tail=G.vae_kl_tail(8, kl_b) if kl_tail else None)]).cpu().numpy()
    finally:
        os.environ.pop("OSRL_NB64", None)
    torch.cuda.synchronize()
    idx0 = np.arange(rows) % B
    X = np.concatenate([src0.cpu().numpy().astype(np.float64)[idx0], src1.cpu().numpy().astype(np.float64)], 1)
    for e in range(E):
        h = X
        for l, a in enumerate(acts):
            h = _act64(a, h @ Ws[e][l][0].T + Ws[e][l][1])
        h = h * oscale
        scale = max(1.0, np.abs(h).max())
        assert np.abs(got[e].reshape(h.shape) - h).max() < 3e-5 * scale, f"case {ci} net {e}: shared-row form vs fp64"
        d = np.abs(got[e] - plain[e]).max()
        assert d < 2e-6 * scale, f"case {ci} net {e}: shared-row form vs plain launch {d:.3e}"
    if B % 16:
        assert np.array_equal(got, plain), "a shape that does not tile runs the plain form"
    if kl_tail:
        ka, kb = kl_a.cpu().numpy(), kl_b.cpu().numpy()
        assert np.isfinite(kb).all() and np.abs(ka - kb).max() < 1e-5 * max(1.0, np.abs(ka).max())
    # the tile kernels ignore the hint: same bits as their plain launch
    desc16 = NetDesc(refs, acts, oscale)
    desc16.c.tile_rows = 16
    run16 = MlpRun(desc16, rows, False, dev)
    y0 = torch.stack([t.clone() for t in run16.forward(src0, src1, map0=1, div0=B)])
    y1 = torch.stack([t.clone() for t in run16.forward(src0, src1, map0=1, div0=B, share_k16=k16)])
    assert torch.equal(y0, y1)


def test_mlp_forward_pair_equals_two_launches():
    """osrl_mlp_forward2 (two independent problems, one launch) == two osrl_mlp_forward calls, incl. saved
    activations, different row counts / net counts / input widths, and the fallback for unequal tile shapes."""
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
    dev = _dev()
    rs = np.random.RandomState(77)

    def mk(E, dims, acts, name):
        grp = FlatGroup(name, dev)
        for e in range(E):
            for l in range(len(dims) - 1):
                grp.add(f"{e}.{l}.w", (dims[l + 1], dims[l]))
                grp.mark_weight(f"{e}.{l}.w")
                grp.add(f"{e}.{l}.b", (dims[l + 1],))
        grp.finalize()
        refs = []
        for e in range(E):
            rr = []
            for l in range(len(dims) - 1):
                W, b = grp.view(f"{e}.{l}.w"), grp.view(f"{e}.{l}.b")
                W.copy_(torch.tensor(rs.uniform(-0.2, 0.2, W.shape), dtype=torch.float32))
                b.copy_(torch.tensor(rs.uniform(-0.2, 0.2, b.shape), dtype=torch.float32))
                rr.append(LayerRef(W, b, grp, f"{e}.{l}.w", f"{e}.{l}.b"))
            refs.append(rr)
        grp.repack()
        return grp, NetDesc(refs, acts, 1.0)

    cases = [((1, [76, 256, 256, 4], 2048, True), (1, [76, 256, 256, 4], 2048, False)),      # actor on obs / next_obs
             ((2, [78, 256, 256, 1], 2048, False), (2, [78, 256, 256, 1], 2048, True)),      # target + live cost critics
             ((4, [78, 256, 256, 1], 1000, False), (2, [78, 256, 256, 1], 2048 + 7, True)),  # ragged, unequal rows
             ((1, [78, 400, 400, 8], 300, True), (1, [80, 400, 400, 2], 500, False)),        # 7-block shape
             ((1, [78, 400, 400, 8], 300, False), (2, [78, 256, 256, 1], 300, False))]       # shapes differ: fallback
    for (E0, d0, r0, s0), (E1, d1, r1, s1) in cases:
        g0, n0 = mk(E0, d0, ["relu", "relu", "id"], "a")
        g1, n1 = mk(E1, d1, ["relu", "relu", "id"], "b")
        x0, x1 = torch.randn(r0, d0[0], device=dev), torch.randn(r1, d1[0], device=dev)
        ref0, ref1 = MlpRun(n0, r0, s0, dev), MlpRun(n1, r1, s1, dev)
        ref0.forward(x0)
        ref1.forward(x1)
        p0, p1 = MlpRun(n0, r0, s0, dev), MlpRun(n1, r1, s1, dev)
        y0, y1 = p0.forward_with((x0,), p1, (x1,))
        torch.cuda.synchronize()
        # (each workgroup starts its k-walk at a step derived from its block index, so the fp32 summation order of
        # the second problem differs from a stand-alone launch: equal to round-off, not bit-equal)
        close = lambda u, v: (u - v).abs().max().item() <= 2e-6 * max(1.0, v.abs().max().item())  # noqa: E731
        for a, b in ((ref0, p0), (ref1, p1)):
            assert close(a.y, b.y)
            if a.save:
                assert torch.equal(a.x, b.x)
                for e in range(a.net.E):
                    for l in range(a.net.nl):
                        assert close(a.h[e][l], b.h[e][l])


@pytest.mark.parametrize("rows,od,ad,hid,tile", [(2048, 76, 2, 400, 0), (300, 17, 6, 256, 0), (37, 5, 3, 64, 0),
                                                 (4096 + 5, 20, 4, 400, 0), (20480, 76, 2, 400, 80)])
def test_vae_tails_fused_into_the_mlp_launches(rows, od, ad, hid, tile):
    """osrl_mlp_forward_tail / osrl_mlp_backward_dz_tail (reparameterisation and its backward applied to the launch's
    LDS-resident last tile) == the plain launch followed by osrl_vae_latent / osrl_vae_latent_bwd, bit for bit; also
    where the library has to fall back to two launches (the 80-row forward, wide dX slices)."""
    from osrl_amd.engine import glue as G
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
    dev = _dev()
    rs = np.random.RandomState(rows + ad)
    Lz = 2 * ad

    def mk(dims, acts, name):
        grp = FlatGroup(name, dev)
        for l in range(len(dims) - 1):
            grp.add(f"{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{l}.w")
            grp.add(f"{l}.b", (dims[l + 1],))
        grp.finalize()
        rr = []
        for l in range(len(dims) - 1):
            W, b = grp.view(f"{l}.w"), grp.view(f"{l}.b")
            W.copy_(torch.tensor(rs.uniform(-0.2, 0.2, W.shape), dtype=torch.float32))
            b.copy_(torch.tensor(rs.uniform(-0.2, 0.2, b.shape), dtype=torch.float32))
            rr.append(LayerRef(W, b, grp, f"{l}.w", f"{l}.b"))
        grp.repack()
        return grp, NetDesc([rr], acts, 1.0)

    _, enc = mk([od + ad, hid, hid, 2 * Lz], ["relu", "relu", "id"], "enc")
    _, dec = mk([od + Lz, hid, hid, ad], ["relu", "relu", "tanh"], "dec")
    obs, act = torch.randn(rows, od, device=dev), torch.rand(rows, ad, device=dev) * 2 - 1
    eps = torch.randn(rows, Lz, device=dev)
    # forward
    r_a, r_b = MlpRun(enc, rows, tile == 0, dev, tile_rows=tile), MlpRun(enc, rows, tile == 0, dev, tile_rows=tile)
    z_a, z_b = torch.full((rows, Lz), 7.0, device=dev), torch.full((rows, Lz), -7.0, device=dev)
    head_a = r_a.forward(obs, act)[0]
    G.vae_latent(head_a, eps, rows, Lz, z_a)
    head_b = r_b.forward(obs, act, tail=G.vae_latent_tail(eps, Lz, z_b))[0]
    torch.cuda.synchronize()
    assert torch.equal(head_a, head_b) and torch.equal(z_a, z_b)
    if tile:
        return
    # backward of the decoder through z
    d_a, d_b = MlpRun(dec, rows, True, dev), MlpRun(dec, rows, True, dev)
    du = torch.randn(1, rows, ad, device=dev)
    for d in (d_a, d_b):
        d.forward(obs, z_a)
        d.setup_backward(du, need_dz=True, dx_cols=(od, Lz))
    dh_a, dh_b = torch.full((rows, 2 * Lz), 3.0, device=dev), torch.full((rows, 2 * Lz), -3.0, device=dev)
    d_a.backward_dz()
    G.vae_latent_bwd(head_a, eps, d_a.dx, rows, Lz, 0.5, rows + 11, dh_a)
    d_b.backward_dz(tail=G.vae_latent_bwd_tail(head_a, eps, Lz, 0.5, rows + 11, dh_b))
    torch.cuda.synchronize()
    assert torch.equal(d_a.dx, d_b.dx) and torch.equal(dh_a, dh_b)
    for l in range(3):
        assert torch.equal(d_a.dz[0][l], d_b.dz[0][l])


@pytest.mark.parametrize("rows,od,ad,hid,N", [(2048, 76, 2, 256, 10), (2048, 17, 6, 256, 10), (37, 5, 3, 64, 4),
                                              (300, 20, 4, 96, 3)])
def test_gauss_head_tails_fused_into_the_actor_forward_pair(rows, od, ad, hid, N):
    """osrl_mlp_forward2_tail with OSRL_TAIL_GAUSS tails (the CPQ step's four action draws made by the actor trunks' own
    paired forward launch) == osrl_mlp_forward2 followed by osrl_gauss_head x3 + osrl_gauss_ood_sample, bit for bit."""
    from osrl_amd.engine import glue as G
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
    dev = _dev()
    rs = np.random.RandomState(rows + ad)
    grp = FlatGroup("actor", dev)
    dims, acts = [od, hid, hid, 2 * ad], ["relu", "relu", "id"]
    for l in range(3):
        grp.add(f"{l}.w", (dims[l + 1], dims[l]))
        grp.mark_weight(f"{l}.w")
        grp.add(f"{l}.b", (dims[l + 1],))
    grp.finalize()
    rr = []
    for l in range(3):
        W, b = grp.view(f"{l}.w"), grp.view(f"{l}.b")
        W.copy_(torch.tensor(rs.uniform(-0.3, 0.3, W.shape), dtype=torch.float32))
        b.copy_(torch.tensor(rs.uniform(-0.3, 0.3, b.shape), dtype=torch.float32))
        rr.append(LayerRef(W, b, grp, f"{l}.w", f"{l}.b"))
    grp.repack()
    net = NetDesc([rr], acts, 1.0)
    nobs, obs = torch.randn(rows, od, device=dev), torch.randn(rows, od, device=dev)
    e_cc, e_c, e_pi = (torch.randn(rows, ad, device=dev) for _ in range(3))
    e_ood = torch.randn(N, rows, ad, device=dev)
    res = []
    for fused in (False, True):
        r_n, r_o = MlpRun(net, rows, False, dev), MlpRun(net, rows, True, dev)
        fill = 5.0 if fused else -5.0
        a2, a1, api, th = (torch.full((rows, ad), fill, device=dev) for _ in range(4))
        smp = torch.full((N * rows, ad), fill, device=dev)
        if fused:
            hn, ho = r_n.forward_with((nobs,), r_o, (obs,),
                                      tail=G.gauss_tail(ad, 1.5, eps=e_cc, a=a2, eps2=e_c, a2=a1),
                                      other_tail=G.gauss_tail(ad, 1.5, eps2=e_pi, a2=api, tanh2=th, eps_ood=e_ood,
                                                              n_samples=N, sampled=smp))
        else:
            hn, ho = r_n.forward_with((nobs,), r_o, (obs,))
            G.gauss_head(hn[0], e_cc, rows, ad, 1.5, a=a2)
            G.gauss_head(hn[0], e_c, rows, ad, 1.5, a=a1)
            G.gauss_ood_sample(ho[0], e_ood, N, rows, ad, smp)
            G.gauss_head(ho[0], e_pi, rows, ad, 1.5, a=api, tanh_u=th)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (hn[0], ho[0], a2, a1, api, th, smp)])
    for x, y, name in zip(res[0], res[1], ("head_next", "head_obs", "a_next2", "a_next", "a_pi", "tanh_u", "sampled")):
        assert torch.equal(x, y), name


@pytest.mark.parametrize("rows,N,nq", [(4096, 10, 2), (4096 + 37, 7, 1), (300, 3, 2), (70000, 4, 2)])
def test_grid_losses_match_the_single_workgroup_kernels(rows, N, nq):
    """osrl_bcq_critic_loss_ws / osrl_vae_loss_ws (<= 64 workgroups, partials added in workgroup order by the last
    arriver) == the single-workgroup kernels: per-row gradients bit for bit, the logged sums to fp32 summation-order
    round-off; twice in a row on the same scratch (the arrival counter re-arms itself) and deterministic."""
    from osrl_amd.engine import glue as G
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(rows + N)
    r = lambda *s_: torch.randn(*s_, generator=g).to(dev)  # noqa: E731
    q_t, q_on = r(2 * nq, rows * N, 1), r(2 * nq, rows, 1)
    base, done = r(rows), (torch.rand(rows, generator=g) < 0.1).float().to(dev)
    dq_a, dq_b = torch.zeros_like(q_on), torch.zeros_like(q_on)
    st_a, st_b = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
    ws = G.loss_ws(dev)
    G.bcq_critic_loss(q_t, nq, nq, N, q_on, 2 * nq, base, done, rows, 0.99, 0.75, rows + 5, dq_a, st_a)
    vals = []
    for _ in range(3):
        G.bcq_critic_loss(q_t, nq, nq, N, q_on, 2 * nq, base, done, rows, 0.99, 0.75, rows + 5, dq_b, st_b, ws=ws)
        torch.cuda.synchronize()
        vals.append(float(st_b[0]))
    assert torch.equal(dq_a, dq_b)
    assert abs(vals[0] - float(st_a[0])) <= 2e-6 * max(1.0, abs(float(st_a[0]))) and vals[0] == vals[1] == vals[2]
    ad, Lz = 3, 6
    u, act, head = r(rows, ad), r(rows, ad), r(rows, 2 * Lz) * 0.3
    du_a, du_b = torch.zeros(1, rows, ad, device=dev), torch.zeros(1, rows, ad, device=dev)
    ws2 = G.loss_ws(dev)
    G.vae_loss(u, act, head, rows, ad, Lz, 0.5, rows, du_a, st_a)
    vals = []
    for _ in range(3):
        G.vae_loss(u, act, head, rows, ad, Lz, 0.5, rows, du_b, st_b, ws=ws2)
        torch.cuda.synchronize()
        vals.append(float(st_b[0]))
    assert torch.equal(du_a, du_b)
    assert abs(vals[0] - float(st_a[0])) <= 2e-6 * max(1.0, abs(float(st_a[0]))) and vals[0] == vals[1] == vals[2]


def test_adam_polyak_matches_oracle():
    from oracle.osrl_oracle import Adam
    from osrl_amd.engine.core import FlatGroup, StepState
    dev = _dev()
    rs = np.random.RandomState(3)
    grp = FlatGroup("g", dev, with_target=True)
    grp.add("w", (37, 5))
    grp.add("b", (6,))
    grp.finalize()
    p0 = {"w": rs.randn(37, 5).astype(np.float32), "b": rs.randn(6).astype(np.float32)}
    tgt = {k: v.copy() for k, v in p0.items()}
    for k in p0:
        grp.view(k).copy_(torch.tensor(p0[k]))
        grp.tgt_view(k).copy_(torch.tensor(p0[k]))
    grp.ensure_slabs(3)
    grp.cur_splits = 3
    st = StepState(dev, ["x"])
    opt = Adam(["w", "b"], 1e-3)
    p = {k: v.copy() for k, v in p0.items()}
    tau = 0.005
    for step in range(5):
        gs = {k: rs.randn(3, *p0[k].shape).astype(np.float32) for k in p0}
        for k in p0:
            off, shape = grp.layout[k]
            grp.slabs[:3, off:off + p0[k].size] = torch.tensor(gs[k].reshape(3, -1))
        st.tick()
        grp.adam_step(1e-3, st.ptr, tau=tau)
        opt.step(p, {k: gs[k].sum(0) for k in p0})
        for k in p0:
            tgt[k] = (tau * p[k] + (1 - tau) * tgt[k]).astype(np.float32)
    torch.cuda.synchronize()
    assert st.device_step() == 5
    for k in p0:
        np.testing.assert_allclose(grp.view(k).cpu().numpy(), p[k], atol=2e-6, rtol=0)
        np.testing.assert_allclose(grp.tgt_view(k).cpu().numpy(), tgt[k], atol=2e-6, rtol=0)


def test_adam_step_refreshes_packed_copies():
    """osrl_adam_step_packed: after an optimizer step the fragment-ordered copies pf / pb (and tf for the Polyak
    target) equal a fresh osrl_pack_weights of the updated parameters, padding included (odd shapes, unaligned
    row lengths, a packed two-head alias, non-weight tensors in between)."""
    from osrl_amd.engine.core import FlatGroup, StepState
    dev = _dev()
    rs = np.random.RandomState(31)
    grp = FlatGroup("g", dev, with_target=True)
    grp.add("a.w", (37, 5))
    grp.mark_weight("a.w")
    grp.add("a.b", (37,))
    grp.add("mu.w", (3, 50), align=True)
    grp.add("ls.w", (3, 50), align=False)      # adjacent -> one [6, 50] packed head
    grp.alias("head.w", "mu.w", (6, 50))
    grp.mark_weight("head.w")
    grp.add("c.w", (256, 78))
    grp.mark_weight("c.w")
    grp.add("c.b", (256,))
    grp.finalize()
    grp.p.copy_(torch.tensor(rs.randn(grp.n), dtype=torch.float32))
    grp.tgt.copy_(torch.tensor(rs.randn(grp.n), dtype=torch.float32))
    grp.repack()
    grp.ensure_slabs(2)
    grp.cur_splits = 2
    st = StepState(dev, ["x"])
    for _ in range(3):
        grp.slabs.copy_(torch.tensor(rs.randn(2, grp.n), dtype=torch.float32))
        st.tick()
        grp.adam_step(1e-2, st.ptr, tau=0.3)
    got = [t.clone() for t in (grp.pf, grp.pb, grp.tf)]
    grp.repack()
    torch.cuda.synchronize()
    for name, a, b in zip(("pf", "pb", "tf"), got, (grp.pf, grp.pb, grp.tf)):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("n", [1, 7, 1000, 20480, 163840])
def test_quantile_exact(n):
    from osrl_amd.engine import glue as G
    dev = _dev()
    rs = np.random.RandomState(n)
    x = rs.randn(n).astype(np.float32)
    if n > 100:
        x[::7] = x[3]  # duplicates
        x[5] = -0.0
        x[6] = 0.0
    xt = torch.tensor(x, device=dev)
    out = torch.zeros(4, device=dev)
    for q in (0.75, 0.0, 1.0, 0.5):
        G.quantile(xt, n, q, out)
        ref = torch.quantile(torch.tensor(x), q).item()
        assert abs(out[0].item() - ref) <= 1e-6 * max(1, abs(ref)), (n, q, out[0].item(), ref)


@pytest.mark.parametrize("n", [64, 1025, 20480, 32768, 32769])
def test_quantile_clustered_values_both_select_paths(n):
    """KL-like inputs (positive, sharing their high bytes, many exact ties) on both sides of the register / streaming
    switch at n = 32 * 1024; the two order statistics are exact, the interpolation is within one ulp."""
    from osrl_amd.engine import glue as G
    dev = _dev()
    rs = np.random.RandomState(n)
    x = (0.5 + 1e-3 * rs.rand(n)).astype(np.float32)
    x[rs.randint(0, n, n // 3)] = x[0]
    xt = torch.tensor(x, device=dev)
    out = torch.zeros(4, device=dev)
    xs = np.sort(x)
    for q in (0.75, 0.25, 0.999, 0.0, 1.0):
        G.quantile(xt, n, q, out)
        pos = np.float64(np.float32(q)) * (n - 1)
        lo = int(np.floor(pos))
        hi = min(lo + 1, n - 1)
        ref = np.float32(xs[lo] + (xs[hi] - xs[lo]) * np.float32(pos - lo))
        assert abs(out[0].item() - ref) <= np.spacing(ref), (n, q, out[0].item(), ref)  # fma vs mul+add


@pytest.mark.parametrize("n", [1, 33, 40000, 163840, 200001])
def test_quantile_grid_select_equals_single_workgroup(n):
    """osrl_quantile_ws (multi-workgroup radix select: the data-parallel batch-global quantile) returns the bits of
    osrl_quantile on clustered, tie-heavy and signed inputs; its workspace is left zeroed (two calls in a row)."""
    from osrl_amd import _lib as L
    from osrl_amd.engine import glue as G
    dev = _dev()
    rs = np.random.RandomState(n)
    ws = torch.zeros(L.QUANTILE_WS, dtype=torch.int32, device=dev)
    for kind in ("clustered", "normal"):
        x = (0.5 + 1e-3 * rs.rand(n)).astype(np.float32) if kind == "clustered" else rs.randn(n).astype(np.float32)
        if n > 10:
            x[rs.randint(0, n, n // 3)] = x[0]
        xt = torch.tensor(x, device=dev)
        a, b = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
        for q in (0.75, 0.0, 1.0, 0.3333):
            G.quantile(xt, n, q, a)
            G.quantile_ws(xt, n, q, ws, b)
            assert a[0].item() == b[0].item(), (n, kind, q, a[0].item(), b[0].item())
            assert int(ws[:1024].abs().sum().item()) == 0


@pytest.mark.parametrize("N,B,nqc", [(10, 2048, 2), (10, 256, 1), (3, 1000, 2)])
def test_cpq_ood_stat_equals_quantile_then_mean(N, B, nqc):
    """The fused single-GPU launch (quantile + masked OOD mean) returns the bits of the two separate launches."""
    from osrl_amd.engine import glue as G
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(N * B)
    kl = (0.3 + 0.01 * torch.rand(N * B, generator=g)).to(dev)
    qc = torch.randn(nqc, N * B, generator=g).to(dev)
    q0, m0, q1, m1 = (torch.zeros(4, device=dev) for _ in range(4))
    G.quantile(kl, N * B, 0.75, q0)
    G.cpq_ood_mean(qc, nqc, kl, q0, N, B, B, m0)
    G.cpq_ood_stat(qc, nqc, kl, 0.75, N, B, B, q1, m1)
    assert q0[0].item() == q1[0].item() and m0[0].item() == m1[0].item()
    keep = (kl >= q0[0]).view(N, B)
    ref = (keep * qc.min(0).values.view(N, B)).mean(0).mean().item()
    assert abs(m1[0].item() - ref) <= 1e-5 * max(1.0, abs(ref))


@pytest.mark.parametrize("n,ties", [(7, False), (1000, True), (20480, False), (20480, True), (32768, False)])
def test_cpq_ood_select_lists_the_rows_that_reach_the_quantile(n, ties):
    """osrl_cpq_ood_select (cpq.py:183-184 as a row set): the quantile has the bits of osrl_quantile / torch.quantile, the
    list is exactly the ascending indices with kl >= quantile (ties at the quantile included), the count is their number;
    with a given quantile the same list.  osrl_cpq_ood_sum over compacted values == the masked mean of the full rows."""
    from osrl_amd.engine import glue as G
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(n + int(ties))
    kl = 0.3 + 0.01 * torch.rand(n, generator=g)
    if ties:  # a quarter of the values equal: the quantile lands on a run of equal keys
        kl[torch.randperm(n, generator=g)[: n // 2]] = 0.3095
    kl = kl.to(dev)
    q0, q1 = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
    lst, cnt = torch.full((n,), -1, dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
    G.quantile(kl, n, 0.75, q0)
    G.cpq_ood_select(kl, n, 0.75, q1, lst, cnt)
    torch.cuda.synchronize()
    assert q0[0].item() == q1[0].item() == torch.quantile(kl, 0.75).item()
    want = torch.nonzero(kl >= q1[0]).flatten().to(torch.int32)
    c = int(cnt[0].item())
    assert c == want.numel() and c >= n - int(0.75 * (n - 1)) - 1
    assert torch.equal(lst[:c], want) and bool((lst[c:] == -1).all())
    lst2, cnt2 = torch.full((n,), -1, dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
    G.cpq_ood_select(kl, n, 0.75, None, lst2, cnt2, quantile_in=q0)
    assert torch.equal(lst2, lst) and int(cnt2[0].item()) == c
    # the sum over the compacted outputs against the masked mean over all rows (N = 1: n rows of one sample)
    nqc = 2
    qc = torch.randn(nqc, n, generator=g).to(dev)
    sel = torch.zeros(nqc, n, device=dev)
    sel[:, :c] = qc[:, want.long()]
    out, ref = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
    G.cpq_ood_sum(sel, nqc, n, cnt, 1.0 / n, out)
    G.cpq_ood_mean(qc, nqc, kl, q0, 1, n, n, ref)
    torch.cuda.synchronize()
    assert abs(out[0].item() - ref[0].item()) <= 2e-6 * max(1.0, abs(ref[0].item()))


@pytest.mark.parametrize("E,dims,acts,rows,div0,k", [
    (2, [78, 256, 256, 1], ["relu", "relu", "id"], 20480, 2048, 5120),   # C2's target cost critics on a quarter of the rows
    (2, [23, 256, 256, 1], ["relu", "relu", "id"], 20480, 2048, 5133),   # C4's width, a ragged count
    (1, [41, 208, 250, 1], ["tanh", "relu", "id"], 2560, 256, 1),        # one row
    (2, [78, 256, 256, 1], ["relu", "relu", "id"], 2560, 256, 0),        # an empty set: nothing is written
    (3, [40, 256, 1], ["relu", "id"], 4000, 400, 4000),                  # every row, permuted
])
def test_forward_on_a_device_chosen_row_set(E, dims, acts, rows, div0, k):
    """osrl_rows_t.row_list / n_rows_dev: the 80-row inference forward on the rows list[0 .. *count) of the virtual input
    [src0[r % div0] | src1[r]] writes, compacted in list order, the BITS the launch over all rows writes at those rows;
    outputs past the count are not touched; the grid is sized for the capacity and the count is read on the device.
    Entry points that cannot honour a list refuse it (-3)."""
    import ctypes as C
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc, cur_stream
    dev = _dev()
    rs = np.random.RandomState(7 + k)
    grp = FlatGroup("t", dev)
    for e in range(E):
        for l in range(len(dims) - 1):
            grp.add(f"{e}.{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{e}.{l}.w")
            grp.add(f"{e}.{l}.b", (dims[l + 1],))
    grp.finalize()
    refs = []
    for e in range(E):
        rr = []
        for l in range(len(dims) - 1):
            kk = 1 / math.sqrt(dims[l])
            W, b = grp.view(f"{e}.{l}.w"), grp.view(f"{e}.{l}.b")
            W.copy_(torch.tensor(rs.uniform(-kk, kk, W.shape), dtype=torch.float32))
            b.copy_(torch.tensor(rs.uniform(-kk, kk, b.shape), dtype=torch.float32))
            rr.append(LayerRef(W, b, grp, f"{e}.{l}.w", f"{e}.{l}.b"))
        refs.append(rr)
    grp.repack()
    d0 = dims[0] - 2
    src0 = torch.tensor(rs.randn(div0, d0), dtype=torch.float32, device=dev)
    src1 = torch.tensor(rs.randn(rows, 2), dtype=torch.float32, device=dev)
    desc = NetDesc(refs, acts, 1.0)
    desc.c.tile_rows = 80
    full = MlpRun(desc, rows, False, dev)
    y_full = full.forward(src0, src1, map0=L.MAP_MOD, div0=div0).clone()
    perm = torch.tensor(rs.permutation(rows)[:k] if k < rows else rs.permutation(rows), dtype=torch.int32, device=dev)
    lst = torch.zeros(rows, dtype=torch.int32, device=dev)
    lst[:k] = perm if k == rows else torch.sort(perm).values
    cnt = torch.tensor([k, 0, 0, 0], dtype=torch.int32, device=dev)
    sel = MlpRun(desc, rows, False, dev)
    sel.y.fill_(-77.0)
    y_sel = sel.forward(src0, src1, map0=L.MAP_MOD, div0=div0, row_list=lst, n_rows_dev=cnt)
    torch.cuda.synchronize()
    # (not bit for bit: a workgroup starts its weight stream at a position that depends on its tile index, so a row's k
    # order depends on the TILE it sits in -- 1.3e-7 at |y| <= 1; the identity list below is bit-equal)
    want = y_full[:, lst[:k].long()]
    if k:
        assert (y_sel[:, :k] - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())
    assert bool((y_sel[:, k:] == -77.0).all()), "rows past the count must not be written"
    y_again = sel.forward(src0, src1, map0=L.MAP_MOD, div0=div0, row_list=lst, n_rows_dev=cnt).clone()
    torch.cuda.synchronize()
    assert torch.equal(y_again, y_sel), "the same list must give the same bits"
    # a count above the capacity is clamped to it
    cnt[0] = rows + 1000
    lst2 = torch.arange(rows, dtype=torch.int32, device=dev)
    y2 = sel.forward(src0, src1, map0=L.MAP_MOD, div0=div0, row_list=lst2, n_rows_dev=cnt)
    torch.cuda.synchronize()
    assert torch.equal(y2, y_full)
    # refused where no kernel reads the list: 16-row tiles (training form), the paired forward
    small = NetDesc(refs, acts, 1.0)
    small.c.tile_rows = 16
    r16 = MlpRun(small, rows, False, dev)
    with pytest.raises(RuntimeError):
        r16.forward(src0, src1, map0=L.MAP_MOD, div0=div0, row_list=lst2, n_rows_dev=cnt)


def test_polyak_step_alone_equals_the_step_carried_by_adam():
    """osrl_polyak after osrl_adam_step_packed(tgt = NULL) leaves the bits of the one-launch Adam + Polyak: flat targets
    and their packed forward copy (the plan that moves the target update behind its last reader, engine/cpq.py)."""
    from osrl_amd.engine.core import FlatGroup, StepState
    dev = _dev()
    outs = []
    for split in (False, True):
        rs = np.random.RandomState(5)
        grp = FlatGroup("g", dev, with_target=True)
        grp.add("a.w", (256, 78)); grp.mark_weight("a.w"); grp.add("a.b", (256,))
        grp.add("b.w", (1, 256)); grp.mark_weight("b.w"); grp.add("b.b", (1,))
        grp.finalize()
        grp.p.copy_(torch.tensor(rs.randn(grp.n), dtype=torch.float32))
        grp.tgt.copy_(torch.tensor(rs.randn(grp.n), dtype=torch.float32))
        grp.repack()
        grp.ensure_slabs(2)
        grp.cur_splits = 2
        st = StepState(dev, ["x"])
        for _ in range(3):
            grp.slabs.copy_(torch.tensor(rs.randn(2, grp.n), dtype=torch.float32))
            st.tick()
            if split:
                grp.adam_step(1e-2, st.ptr, tau=0.005, polyak=False)
                grp.polyak_step(0.005)
            else:
                grp.adam_step(1e-2, st.ptr, tau=0.005)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (grp.p, grp.m, grp.v, grp.tgt, grp.pf, grp.pb, grp.tf)])
    for name, a, b in zip(("p", "m", "v", "tgt", "pf", "pb", "tf"), *outs):
        assert torch.equal(a, b), name


def test_randn_and_gather():
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import StepState, cur_stream, randn_fill
    import ctypes as C
    dev = _dev()
    st = StepState(dev, ["x"])
    st.tick()
    a = torch.zeros(1 << 20, device=dev)
    randn_fill(a, 1234, 0, st.ptr)
    b = a.clone()
    randn_fill(a, 1234, 0, st.ptr)
    assert torch.equal(a, b), "same (seed, step, stream) must reproduce"
    st.tick()
    randn_fill(a, 1234, 0, st.ptr)
    assert not torch.equal(a, b), "a new step must draw fresh noise"
    x = a.double().cpu().numpy()
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1) < 5e-3
    assert abs((x ** 4).mean() - 3) < 0.05 and abs(np.mean(x > 1.0) - 0.158655) < 2e-3
    # replay gather
    n_rows, B = 5000, 2048
    tabs = [torch.arange(n_rows * w, device=dev, dtype=torch.float32).view(n_rows, w) for w in (76, 2, 1)]
    outs = [torch.zeros(B, w, device=dev) for w in (76, 2, 1)]
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    src = (C.c_void_p * 3)(*[t.data_ptr() for t in tabs])
    dst = (C.c_void_p * 3)(*[t.data_ptr() for t in outs])
    width = (C.c_int32 * 3)(76, 2, 1)
    scale = (C.c_float * 3)(1.0, 1.0, 0.5)
    L.check(L.load().osrl_replay_gather(3, src, dst, width, scale, n_rows, B, idx.data_ptr(), 99, 1, st.ptr,
                                        cur_stream()), "gather")
    torch.cuda.synchronize()
    ii = idx.long()
    assert ii.min() >= 0 and ii.max() < n_rows and len(torch.unique(ii)) > B * 0.7
    assert torch.equal(outs[0], tabs[0][ii]) and torch.equal(outs[1], tabs[1][ii])
    assert torch.equal(outs[2], tabs[2][ii] * 0.5)
    assert abs(ii.float().mean().item() / n_rows - 0.5) < 0.03


@pytest.mark.parametrize("with_noise,with_gather", [(True, True), (True, False), (False, True)])
def test_fused_step_prologue_equals_tick_randn_gather(with_noise, with_gather):
    """osrl_step_begin (one launch) == osrl_step_tick -> osrl_randn_fill -> osrl_replay_gather over several steps:
    same step state bytes, same statistics ring, same noise, same gathered rows; the arrival counter is back at 0."""
    from osrl_amd.common.replay import ReplayStore, synthetic_transitions
    from osrl_amd.engine.core import StepState, randn_fill
    dev = _dev()
    B, od, ad = 1000, 11, 3
    store = ReplayStore(synthetic_transitions(5000, od, ad), dev, reward_scale=0.5, seed=7)
    mk = lambda: [torch.zeros(B, w, device=dev) for w in (od, od, ad, 1, 1, 1)]  # noqa: E731
    sa, sb = StepState(dev, ["x", "y"], ring_len=4), StepState(dev, ["x", "y"], ring_len=4)
    na, nb = torch.zeros(12346, device=dev), torch.zeros(12346, device=dev)
    da, db = mk(), mk()
    for step in range(7):
        for s_ in (sa, sb):
            s_.stats.copy_(torch.tensor([step + 0.5, -step], device=dev))
        sa.tick()
        if with_noise:
            randn_fill(na, 4321, 0, sa.ptr)
        if with_gather:
            store.gather(da, sa.ptr)
        sb.begin(nb if with_noise else None, 4321, 0, store.gather_args(db) if with_gather else None)
        torch.cuda.synchronize()
        assert torch.equal(sa.state, sb.state), step
        assert int(sb.state[20:24].view(torch.int32).item()) == 0
        assert torch.equal(sa.ring, sb.ring) and sa.host_step == sb.host_step == step + 1
        assert torch.equal(na, nb)
        for x, y in zip(da, db):
            assert torch.equal(x, y)
    assert sb.device_step() == 7 and (not with_noise or na.abs().sum() > 0)


def test_autograd_function_matches_torch():
    """osrl_amd.ops.mlp_apply is a torch.autograd.Function over the HIP kernels: forward values and the
    gradients w.r.t. inputs, weights and biases must match torch's own autograd on a CPU fp64 copy."""
    from osrl_amd import ops
    from osrl_amd.algorithms import CPQ
    dev = _dev()
    torch.manual_seed(3)
    m = CPQ(7, 3, 1.0, [48, 32], [48, 32], 40, 3, num_q=2, num_qc=1, device=str(dev))
    obs = torch.randn(37, 7, device=dev, requires_grad=True)
    act = torch.randn(37, 3, device=dev, requires_grad=True)
    for p in m.critic.parameters():
        p.requires_grad_(True)
    q_list = m.critic(obs, act)                      # EnsembleQCritic.forward -> fused HIP MLP
    loss = sum((q * (i + 1.5)).pow(2).mean() for i, q in enumerate(q_list))
    loss.backward()
    # torch reference on CPU in fp64
    ref_nets = []
    for qn in m.critic.q_nets:
        layers = []
        for mod in qn:
            if isinstance(mod, torch.nn.Linear):
                l = torch.nn.Linear(mod.in_features, mod.out_features).double()
                l.weight.data = mod.weight.data.detach().cpu().double().clone()
                l.bias.data = mod.bias.data.detach().cpu().double().clone()
                layers.append(l)
            else:
                layers.append(type(mod)())
        ref_nets.append(torch.nn.Sequential(*layers))
    o2 = obs.detach().cpu().double().requires_grad_(True)
    a2 = act.detach().cpu().double().requires_grad_(True)
    x = torch.cat([o2, a2], 1)
    loss2 = sum((rn(x).squeeze(-1) * (i + 1.5)).pow(2).mean() for i, rn in enumerate(ref_nets))
    loss2.backward()
    assert abs(loss.item() - loss2.item()) < 1e-5 * max(1, abs(loss2.item()))
    assert (obs.grad.cpu().double() - o2.grad).abs().max() < 1e-5
    assert (act.grad.cpu().double() - a2.grad).abs().max() < 1e-5
    for qn, rn in zip(m.critic.q_nets, ref_nets):
        for mod, rmod in zip(qn, rn):
            if isinstance(mod, torch.nn.Linear):
                assert (mod.weight.grad.cpu().double() - rmod.weight.grad).abs().max() < 2e-5
                assert (mod.bias.grad.cpu().double() - rmod.bias.grad).abs().max() < 2e-5


def test_bear_mmd_kernel_random_shapes():
    """osrl_bear_mmd against the fp64 oracle (oracle/bearl_oracle.py mmd_and_grad) at the kernel's limits: M = 1 and
    64 samples, action dims 1..16, both kernels, rows not a multiple of the 4 rows a workgroup handles; plus u,
    tanh(u) and the sample-0 action."""
    from oracle.bearl_oracle import mmd_and_grad
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import cur_stream
    dev = _dev()
    lib = L.load()
    rs = np.random.RandomState(5)
    for (B, M, ad, sigma, kern) in [(1, 1, 1, 0.7, 0), (7, 64, 16, 20.0, 0), (33, 10, 8, 50.0, 0), (5, 3, 2, 1.5, 1),
                                    (130, 10, 3, 2.0, 1), (9, 64, 1, 0.3, 1), (4, 17, 16, 5.0, 0)]:
        x = rs.randn(B * M, ad).astype(np.float32)
        head = np.concatenate([rs.randn(B * M, ad), rs.uniform(-3, 1, (B * M, ad))], 1).astype(np.float32)
        head[0, ad] = 5.0  # log_std above the clamp (2.0)
        eps = rs.randn(B * M, ad).astype(np.float32)
        tt = lambda a: torch.tensor(a, device=dev)  # noqa: E731
        xt, ht, et = tt(x), tt(head), tt(eps)
        mmd, du = torch.zeros(B, device=dev), torch.zeros(B * M, ad, device=dev)
        tu, a0 = torch.zeros(B * M, ad, device=dev), torch.zeros(B, ad, device=dev)
        L.check(lib.osrl_bear_mmd(xt.data_ptr(), ht.data_ptr(), et.data_ptr(), B, M, ad, sigma, kern, mmd.data_ptr(),
                                  du.data_ptr(), tu.data_ptr(), a0.data_ptr(), cur_stream()), "mmd")
        mu, ls = head[:, :ad].astype(np.float64), np.clip(head[:, ad:].astype(np.float64), -20, 2)
        u = mu + np.exp(ls) * eps
        ref, dref = mmd_and_grad(x.astype(np.float64).reshape(B, M, ad), u.reshape(B, M, ad), sigma,
                                 "gaussian" if kern == 0 else "laplacian")
        np.testing.assert_allclose(mmd.cpu().numpy(), ref, rtol=2e-4, atol=2e-6, err_msg=str((B, M, ad, kern)))
        sc = max(1e-3, np.abs(dref).max())
        assert np.abs(du.cpu().numpy().reshape(B, M, ad) - dref).max() < 2e-3 * sc, (B, M, ad, kern)
        np.testing.assert_allclose(tu.cpu().numpy(), np.tanh(u), atol=2e-6)
        np.testing.assert_allclose(a0.cpu().numpy(), np.tanh(u).reshape(B, M, ad)[:, 0], atol=2e-6)
    # limits are enforced
    z = torch.zeros(8, device=dev)
    assert lib.osrl_bear_mmd(z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 65, 1, 1.0, 0, z.data_ptr(), z.data_ptr(),
                             z.data_ptr(), z.data_ptr(), cur_stream()) == -1
    assert lib.osrl_bear_mmd(z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 2, 17, 1.0, 0, z.data_ptr(), z.data_ptr(),
                             z.data_ptr(), z.data_ptr(), cur_stream()) == -1


@pytest.mark.parametrize("B,n_chi,scale", [(1, 1, 1.0), (37, 3, 1.0), (3000, 2, 1.0), (2500, 2, 40.0)])
def test_dice_chi_step_kernel(B, n_chi, scale):
    """osrl_dice_chi_step vs fp64 numpy: the batch softmax (also with logits of magnitude ~1e3: max-subtraction),
    D_kl, weighted_c, chi_loss, its gradient through the NOT-detached weights with the min-routing of predict, and
    the Adam step on tau; B beyond one pass of the 1024-thread workgroup."""
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import StepState, cur_stream
    dev = _dev()
    lib = L.load()
    rs = np.random.RandomState(B)
    gamma, p0, eps_ub, lr = 0.99, 0.07, 0.01, 1e-2
    chi2 = (rs.randn(n_chi, 2 * B) * scale).astype(np.float32)
    w = np.abs(rs.randn(B)).astype(np.float32)
    cost = (rs.uniform(size=B) < 0.3).astype(np.float32)
    done = (rs.uniform(size=B) < 0.1).astype(np.float32)
    init = (rs.uniform(size=B) < 0.2).astype(np.float32)
    tt = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    leaves = tt(np.array([0.4, 0, 0, 1.0, 0, 0], np.float32))
    tau_p = float(np.logaddexp(0.0, 0.4))
    work = tt(np.array([1.3, tau_p, 0, 0], np.float32))
    st = StepState(dev, ["a", "b", "c"])
    st.tick()
    ell_ws, dchi = torch.zeros(B, device=dev), torch.zeros(n_chi, 2 * B, device=dev)
    c2, wt, ct, dt_, it = tt(chi2), tt(w), tt(cost), tt(done), tt(init)
    L.check(lib.osrl_dice_chi_step(c2.data_ptr(), n_chi, B, wt.data_ptr(), ct.data_ptr(), dt_.data_ptr(), it.data_ptr(),
                                   gamma, p0, eps_ub, lr, st.ptr, leaves.data_ptr(), work.data_ptr(), ell_ws.data_ptr(),
                                   dchi.data_ptr(), None, 0, 0, 1.0, st.stats.data_ptr(), cur_stream()), "chi")
    if B % 2 == 0 or B > 100:
        # data-parallel form: two "ranks" hold the halves, ell is gathered, every rank sees the global statistics
        # (x share 1/2) and the gradient of its own rows
        h = B // 2
        parts = [(0, h), (h, B - h)]
        ells = []
        keep = []
        for r0, n in parts:
            sub = np.concatenate([chi2[:, r0:r0 + n], chi2[:, B + r0:B + r0 + n]], 1).copy()
            e_r = torch.zeros(n, device=dev)
            args = (tt(sub), tt(w[r0:r0 + n]), tt(cost[r0:r0 + n]), tt(done[r0:r0 + n]), tt(init[r0:r0 + n]))
            L.check(lib.osrl_dice_chi_ell(args[0].data_ptr(), n_chi, n, args[1].data_ptr(), args[2].data_ptr(),
                                          args[3].data_ptr(), args[4].data_ptr(), gamma, p0, e_r.data_ptr(),
                                          cur_stream()), "ell")
            ells.append(e_r)
            keep.append(args)
        ell_all = torch.cat(ells)
        wc_sum, stats_sum = 0.0, np.zeros(3)
        for (r0, n), args, e_r in zip(parts, keep, ells):
            lv = tt(np.array([0.4, 0, 0, 1.0, 0, 0], np.float32))
            wk = tt(np.array([1.3, tau_p, 0, 0], np.float32))
            st2 = StepState(dev, ["a", "b", "c"])
            st2.tick()
            dc = torch.zeros(n_chi, 2 * n, device=dev)
            L.check(lib.osrl_dice_chi_step(args[0].data_ptr(), n_chi, n, args[1].data_ptr(), args[2].data_ptr(),
                                           args[3].data_ptr(), args[4].data_ptr(), gamma, p0, eps_ub, lr, st2.ptr,
                                           lv.data_ptr(), wk.data_ptr(), e_r.data_ptr(), dc.data_ptr(), ell_all.data_ptr(),
                                           B, r0, 0.5, st2.stats.data_ptr(), cur_stream()), "chi dp")
            wc_sum += wk[2].item()
            stats_sum += st2.stats.cpu().numpy()
            full = dchi.cpu().numpy()
            np.testing.assert_allclose(dc.cpu().numpy()[:, :n], full[:, r0:r0 + n], rtol=2e-4, atol=1e-7 * max(1, scale))
            np.testing.assert_allclose(dc.cpu().numpy()[:, n:], full[:, B + r0:B + r0 + n], rtol=2e-4, atol=1e-7 * max(1, scale))
            assert abs(lv[0].item() - leaves[0].item()) < 1e-6  # every rank steps tau identically
        np.testing.assert_allclose(stats_sum, st.stats.cpu().numpy(), rtol=2e-4, atol=1e-6)
        assert abs(wc_sum - work[2].item()) <= 2e-4 * max(1.0, abs(work[2].item()))
    c64 = chi2.astype(np.float64)
    i_s, i_n = c64[:, :B].argmin(0), c64[:, B:].argmin(0)
    cs, cn = c64[:, :B].min(0), c64[:, B:].min(0)
    ell = (1 - gamma) * cs * init / p0 + w * (cost + gamma * (1 - done) * cn - cs)
    z = ell / tau_p - (ell / tau_p).max()
    lsm = z - np.log(np.exp(z).sum())
    sm = np.exp(lsm)
    wts = sm * B
    dkl = (wts * (lsm + np.log(B)) - wts + 1).mean()
    wc, cl = (wts * w * cost).mean(), (wts * ell).mean()
    dl = sm * (1 + (ell - cl) / tau_p)
    ref = np.zeros((n_chi, 2 * B))
    ref[i_s, np.arange(B)] = dl * ((1 - gamma) * init / p0 - w)
    ref[i_n, B + np.arange(B)] = dl * w * gamma * (1 - done)
    got = st.stats.cpu().numpy()
    tol = 2e-4 if scale > 1 else 2e-5
    assert abs(got[0] - cl) <= tol * max(1, abs(cl)) and abs(got[2] - dkl) <= tol * max(1, abs(dkl)), (got, cl, dkl)
    assert abs(got[1] - tau_p * (eps_ub - dkl)) <= tol * max(1, abs(dkl) * tau_p)
    assert abs(work[2].item() - wc) <= tol * max(1, abs(wc))
    np.testing.assert_allclose(dchi.cpu().numpy(), ref, rtol=5e-4 if scale > 1 else 5e-5, atol=1e-7 * max(1, scale))
    g = (1 / (1 + np.exp(-0.4))) * (eps_ub - dkl)  # first Adam step: p -= lr * sign-like m/(sqrt(v)+eps)
    want = 0.4 - lr * g / (abs(g) + 1e-8)
    assert abs(leaves[0].item() - want) < 1e-5, (leaves[0].item(), want)


@pytest.mark.parametrize("rows,out_f,in_f,force_big", [(8192 + 40, 256, 128, None), (3000, 128, 192, True),
                                                       (16384, 384, 256, None), (8200, 128, 64, None)])
def test_dw_big_rows_kernel(rows, out_f, in_f, force_big):
    """osrl_mlp_backward_dw_big (one wave per 128x64 tile, one wave per SIMD; CDT's token-matrix dW) vs fp64, next to a
    ragged layer of the same plan that stays with the 64x64-tile kernel; ragged row counts and several row splits."""
    from osrl_amd.engine.core import DwPlan, FlatGroup
    dev = _dev()
    rs = np.random.RandomState(rows % 97)
    grp = FlatGroup("t", dev)
    for k, shp in (("a.w", (out_f, in_f)), ("a.b", (out_f,)), ("s.w", (40, 24)), ("s.b", (40,))):
        grp.add(k, shp)
    grp.finalize()
    dz = torch.tensor(rs.randn(rows, out_f), dtype=torch.float32, device=dev)
    a = torch.tensor(rs.randn(rows, in_f), dtype=torch.float32, device=dev)
    dz2 = torch.tensor(rs.randn(rows, 40), dtype=torch.float32, device=dev)
    a2 = torch.tensor(rs.randn(rows, 24), dtype=torch.float32, device=dev)
    plan = DwPlan(grp, [(dz, a, "a.w", "a.b"), (dz2, a2, "s.w", "s.b")], rows, dev, big=force_big)
    assert plan.n_big == (out_f // 128) * (in_f // 64) and plan.n_items > 0
    plan.launch()
    torch.cuda.synchronize()
    for (z, x, wk, bk) in ((dz, a, "a.w", "a.b"), (dz2, a2, "s.w", "s.b")):
        want = z.cpu().numpy().astype(np.float64).T @ x.cpu().numpy().astype(np.float64)
        got = grp.grad_view(wk).cpu().numpy()
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (wk, np.abs(got - want).max())
        wb = z.cpu().numpy().astype(np.float64).sum(0)
        assert np.abs(grp.grad_view(bk).cpu().numpy() - wb).max() <= 2e-5 * max(1.0, np.abs(wb).max()), bk


@pytest.mark.parametrize("rows,out_f,in_f", [(16384, 512, 256), (8192 + 16 * 5, 256, 768), (40960, 256, 256)])
def test_dw_workgroup_tile_kernel(rows, out_f, in_f):
    """osrl_mlp_backward_dw_coop (one 8-wave workgroup per 256x256 tile and row split, operands shared through LDS) vs
    fp64, next to a 384-wide layer that stays with the one-wave-per-tile kernel and a ragged one with the 64x64 tiles;
    a row count that is not a multiple of 16 sends the same layer back to the one-wave-per-tile kernel."""
    from osrl_amd.engine.core import DwPlan, FlatGroup
    dev = _dev()
    rs = np.random.RandomState(rows % 89)
    grp = FlatGroup("t", dev)
    for k, shp in (("a.w", (out_f, in_f)), ("a.b", (out_f,)), ("m.w", (384, 128)), ("m.b", (384,)), ("s.w", (40, 24)),
                   ("s.b", (40,))):
        grp.add(k, shp)
    grp.finalize()
    mk = lambda n: torch.tensor(rs.randn(rows, n), dtype=torch.float32, device=dev)  # noqa: E731
    ents = [(mk(out_f), mk(in_f), "a.w", "a.b"), (mk(384), mk(128), "m.w", "m.b"), (mk(40), mk(24), "s.w", "s.b")]
    plan = DwPlan(grp, ents, rows, dev)
    assert plan.n_coop == (out_f // 256) * (in_f // 256) and plan.n_big == 3 * 2 and plan.n_items > 0
    plan.launch()
    torch.cuda.synchronize()
    for (z, x, wk, bk) in ents:
        want = z.cpu().numpy().astype(np.float64).T @ x.cpu().numpy().astype(np.float64)
        got = grp.grad_view(wk).cpu().numpy()
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (wk, np.abs(got - want).max())
        wb = z.cpu().numpy().astype(np.float64).sum(0)
        assert np.abs(grp.grad_view(bk).cpu().numpy() - wb).max() <= 2e-5 * max(1.0, np.abs(wb).max()), bk
    odd = DwPlan(grp, [(e[0][:rows - 3], e[1][:rows - 3]) + e[2:] for e in ents], rows - 3, dev)
    assert odd.n_coop == 0 and odd.n_big == (out_f // 128) * (in_f // 64) + 6


@pytest.mark.parametrize("rows,T,splits,target", [(2048, 5, 3, True), (2048, 4, None, True), (300, 4, 1, False),
                                                  (1000, 2, 4, True), (2048 + 37, 3, 2, False), (4096, 4, 12, True)])
def test_dw_tiles_adam_one_launch_equals_dw_then_adam(rows, T, splits, target):
    """osrl_mlp_backward_dw_tiles_adam (the last split of a tile to arrive applies the optimizer step to it) vs
    osrl_mlp_backward_dw_tiles + osrl_adam_step_packed on a twin group: parameters, moments, Polyak targets and all
    three packed copies BIT-equal after every one of four steps (the arrival counters re-arm themselves); one split
    (no slab, gradient straight from LDS), several, and more than the eight the epilogue keeps in flight; ragged
    tiles, a packed two-head alias, biases."""
    from osrl_amd.engine.core import DwPlan, FlatGroup, StepState
    dev = _dev()
    rs = np.random.RandomState(rows + T)
    shapes = [("a", 400, 78), ("b", 400, 400), ("h", 6, 400)] if T == 5 else [("a", 256, 41), ("b", 100, 256), ("h", 6, 100)]
    groups = []
    for _ in range(2):
        grp = FlatGroup("g", dev, with_target=target)
        for k, o, i in shapes:
            if k == "h":  # two adjacent [3, in] heads packed as one [6, in] weight
                grp.add("mu.w", (3, i), align=True)
                grp.add("ls.w", (3, i), align=False)
                grp.alias("h.w", "mu.w", (6, i))
                grp.add("mu.b", (3,), align=True)
                grp.add("ls.b", (3,), align=False)
                grp.alias("h.b", "mu.b", (6,))
            else:
                grp.add(k + ".w", (o, i))
                grp.add(k + ".b", (o,))
            grp.mark_weight(k + ".w")
        grp.finalize()
        groups.append(grp)
    live = np.zeros(groups[0].n, np.float32)  # alignment padding stays zero, as in every engine: the streaming Adam
    for key, (off, shape) in groups[0].layout.items():  # kernel walks (and Polyak-averages) the padding floats too
        live[off:off + int(np.prod(shape))] = 1.0
    p0 = torch.tensor(rs.randn(groups[0].n).astype(np.float32) * live, device=dev)
    t0 = torch.tensor(rs.randn(groups[0].n).astype(np.float32) * live, device=dev)
    for grp in groups:
        grp.p.copy_(p0)
        if target:
            grp.tgt.copy_(t0)
        grp.repack()
    acts = [(torch.zeros(rows, o, device=dev), torch.zeros(rows, i, device=dev)) for _, o, i in shapes]
    ents = [(dz, a, k + ".w", k + ".b") for (dz, a), (k, _, _) in zip(acts, shapes)]
    plans = [DwPlan(grp, ents, rows, dev, n_splits=splits, tile_blocks=T) for grp in groups]
    assert plans[0].n_work > 0 and not plans[0].n_items and not plans[0].n_big  # the whole plan is the flat work list
    st = StepState(dev, ["x"])
    ga, gb = groups
    for step in range(4):
        for dz, a in acts:
            dz.copy_(torch.tensor(rs.randn(*dz.shape), dtype=torch.float32))
            a.copy_(torch.tensor(rs.randn(*a.shape), dtype=torch.float32))
        st.tick()
        plans[0].launch()
        ga.adam_step(3e-3, st.ptr, tau=0.25)
        plans[1].launch_adam(3e-3, st.ptr, tau=0.25)
        torch.cuda.synchronize()
        assert int(plans[1].d_counters.abs().sum().item()) == 0, "arrival counters not re-armed"
        names = ["p", "m", "v", "pf", "pb"] + (["tgt", "tf"] if target else [])
        for n in names:
            x, y = getattr(ga, n), getattr(gb, n)
            assert torch.equal(x, y), (step, n, float((x - y).abs().max()))
    assert float((ga.p - p0).abs().max()) > 1e-3  # the steps moved the parameters


@pytest.mark.parametrize("rows,od,ad,hid", [(48, 7, 3, 40), (2048, 76, 2, 400)])
def test_forward2_with_a_kl_tail_writes_the_kl_rows(rows, od, ad, hid):
    """osrl_mlp_forward2_tail with an OSRL_TAIL_VAE_KL tail on one problem (the paired kernel does not act on
    that tail kind -- the call must take the two-launch path and still write ``tail->out``) == the encoder forward
    followed by osrl_vae_kl_rows, bit for bit; the partner problem's result is untouched by the choice."""
    from osrl_amd.engine import glue as G
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
    dev = _dev()
    rs = np.random.RandomState(rows + ad)
    Lz = 2 * ad

    def mk(dims, acts, name):
        grp = FlatGroup(name, dev)
        for l in range(len(dims) - 1):
            grp.add(f"{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{l}.w")
            grp.add(f"{l}.b", (dims[l + 1],))
        grp.finalize()
        rr = []
        for l in range(len(dims) - 1):
            W, b = grp.view(f"{l}.w"), grp.view(f"{l}.b")
            W.copy_(torch.tensor(rs.uniform(-0.2, 0.2, W.shape), dtype=torch.float32))
            b.copy_(torch.tensor(rs.uniform(-0.2, 0.2, b.shape), dtype=torch.float32))
            rr.append(LayerRef(W, b, grp, f"{l}.w", f"{l}.b"))
        grp.repack()
        return grp, NetDesc([rr], acts, 1.0)

    _, enc = mk([od + ad, hid, hid, 2 * Lz], ["relu", "relu", "id"], "enc")
    _, dec = mk([od + Lz, hid, hid, ad], ["relu", "relu", "tanh"], "dec")  # same tile shape as enc: a pairable partner
    obs, act, z = torch.randn(rows, od, device=dev), torch.rand(rows, ad, device=dev) * 2 - 1, torch.randn(rows, Lz, device=dev)
    r_e0, r_d0 = MlpRun(enc, rows, False, dev), MlpRun(dec, rows, False, dev)
    r_e1, r_d1 = MlpRun(enc, rows, False, dev), MlpRun(dec, rows, False, dev)
    kl0, kl1 = torch.full((rows,), 7.0, device=dev), torch.full((rows,), -7.0, device=dev)
    h0, u0 = r_e0.forward_with((obs, act), r_d0, (obs, z))
    G.vae_kl_rows(h0[0], rows, Lz, kl0)
    h1, u1 = r_e1.forward_with((obs, act), r_d1, (obs, z), tail=G.vae_kl_tail(Lz, kl1))
    torch.cuda.synchronize()
    # (the partner problem may run on another kernel form in the two-launch path: same sums in another order)
    assert torch.allclose(h0, h1, rtol=1e-5, atol=2e-5) and torch.allclose(u0, u1, rtol=1e-5, atol=2e-5)
    G.vae_kl_rows(h1[0], rows, Lz, kl0)  # the reference rows from the SAME head values
    torch.cuda.synchronize()
    assert torch.equal(kl0, kl1), "the KL tail of a paired forward was not applied"
    # the tail on the SECOND problem of the pair
    kl2 = torch.full((rows,), 3.0, device=dev)
    r_d1.forward_with((obs, z), r_e1, (obs, act), other_tail=G.vae_kl_tail(Lz, kl2))
    torch.cuda.synchronize()
    assert torch.allclose(kl0, kl2, rtol=1e-4, atol=2e-5) and float(kl2.min()) != 3.0


@pytest.mark.parametrize("rows,od,ad,hid,rg", [(100, 7, 3, 80, 0), (64, 76, 2, 400, 0), (2048, 17, 6, 400, 16384),
                                                (1000, 33, 8, 400, 0), (2048, 76, 2, 160, 0), (1536, 20, 4, 240, 0),
                                                (2048, 11, 3, 320, 4096)])
def test_vae_ns_launches_equal_the_fused_launches(rows, od, ad, hid, rg):
    """osrl_vae_ns_forward / _backward (csrc/vae_ns.hip: the VAE phase as five all-CU layer launches) fill the SAME buffers as
    the four fused launches they replace -- forward_tail(enc, VAE_LATENT), forward(dec), backward_dz_seed(dec, MSE + KL
    statistic, VAE_LATENT_BWD tail), backward_dz(enc) -- up to fp32 summation order: every saved activation, z, every dZ
    the dW plan reads, the logged loss.  Shapes: ragged rows / odd dims / one column group; C2's widths; C4's with the
    8-GPU job's rows_global; a latent that straddles two k-steps of the decoder's first layer (33 + 16); two, three and four
    column groups (H = 160 / 240 / 320: ADVICE r5)."""
    from osrl_amd.engine import glue as G
    from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
    dev = _dev()
    rs = np.random.RandomState(rows + ad)
    Lz, beta = 2 * ad, 0.5

    def mk(dims, acts, name, scale):
        grp = FlatGroup(name, dev)
        for l in range(len(dims) - 1):
            grp.add(f"{l}.w", (dims[l + 1], dims[l]))
            grp.mark_weight(f"{l}.w")
            grp.add(f"{l}.b", (dims[l + 1],))
        grp.finalize()
        rr = []
        for l in range(len(dims) - 1):
            W, b = grp.view(f"{l}.w"), grp.view(f"{l}.b")
            k = 1.0 / np.sqrt(dims[l])
            W.copy_(torch.tensor(rs.uniform(-k, k, W.shape), dtype=torch.float32))
            b.copy_(torch.tensor(rs.uniform(-k, k, b.shape), dtype=torch.float32))
            rr.append(LayerRef(W, b, grp, f"{l}.w", f"{l}.b"))
        grp.repack()
        return grp, NetDesc([rr], acts, scale)

    _, enc = mk([od + ad, hid, hid, 2 * Lz], ["relu", "relu", "id"], "enc", 1.0)
    _, dec = mk([od + Lz, hid, hid, ad], ["relu", "relu", "tanh"], "dec", 1.5)
    obs, act = torch.randn(rows, od, device=dev), (torch.rand(rows, ad, device=dev) * 2 - 1) * 1.5
    eps = torch.randn(rows, Lz, device=dev)
    res = []
    for ns in (False, True):
        r_e, r_d = MlpRun(enc, rows, True, dev), MlpRun(dec, rows, True, dev)
        z, du, dhead = torch.zeros(rows, Lz, device=dev), torch.zeros(1, rows, ad, device=dev), torch.zeros(1, rows, 2 * Lz, device=dev)
        stat = torch.zeros(4, device=dev)
        r_d.setup_backward(du, dx_cols=(od, Lz))
        r_e.setup_backward(dhead)
        if ns:
            v = G.VaeNs.build(r_e, r_d, obs, act, eps, z, Lz, beta, rg, stat)
            assert v is not None, "the library must take this shape"
            for _ in range(2):  # twice: the re-armed arrival counter gives the same statistic again
                stat.zero_()
                v.forward()
                v.backward()
        else:
            head = G.vae_encode(r_e, obs, act, eps, Lz, z)
            r_d.forward(obs, z)
            r_d.backward_dz(tail=G.vae_latent_bwd_tail(head, eps, Lz, beta, rg, dhead),
                            seed=G.seed_vae(act, r_e.y[0], rows, ad, Lz, beta, rg, G.SeedStat(dev, 1, rows), stat))
            r_e.backward_dz()
        torch.cuda.synchronize()
        res.append({"enc.x": r_e.x, "enc.h0": r_e.h[0][0], "enc.h1": r_e.h[0][1], "head": r_e.y[0], "z": z, "dec.x": r_d.x,
                    "dec.h0": r_d.h[0][0], "dec.h1": r_d.h[0][1], "u": r_d.y[0], "dec.dz2": r_d.dz[0][2],
                    "dec.dz1": r_d.dz[0][1], "dec.dz0": r_d.dz[0][0], "enc.dz2": r_e.dz[0][2], "enc.dz1": r_e.dz[0][1],
                    "enc.dz0": r_e.dz[0][0], "loss": stat[:1]})
    for k, a in res[0].items():
        b = res[1][k]
        assert torch.isfinite(b).all(), k
        scale = float(a.abs().max()) + 1e-30
        diff = (a - b).abs()
        if k in ("dec.dz1", "dec.dz0", "enc.dz2", "enc.dz1", "enc.dz0"):
            # relu' of a unit within an ulp of zero may fall on either side in the two summation orders: a few elements
            # may differ by a whole term; everything else to round-off
            n_bad = int((diff > 2e-5 * scale).sum())
            assert n_bad <= max(2, a.numel() // 2000), (k, n_bad, float(diff.max()), scale)
        else:
            assert float(diff.max()) <= 2e-5 * scale, (k, float(diff.max()), scale)
