"""GPU parity of the CDT kernels and the CDT train step (osrl_amd, through the C ABI) against numpy
references, the reference-generated goldens and the CDT oracle."""
import math

import numpy as np
import pytest
import torch

from cases import CDT_CASES, CDTCase, make_cdt_batch, make_cdt_params
from oracle_util import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(x):
    return torch.as_tensor(x, device=DEV)


def test_linear_kernel_shapes():
    """osrl_linear on packed weights: forward pack, backward pack, strided A / Y, residual, column groups."""
    import ctypes as C
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import FlatGroup, cur_stream
    rs = np.random.RandomState(0)
    r16 = lambda x: (x + 15) // 16 * 16  # noqa: E731
    for (M, K, N) in [(200, 256, 768), (96, 1024, 256), (130, 256, 1024), (77, 16, 6), (64, 128, 2), (50, 3, 40),
                      (300, 768, 256), (33, 6, 128),
                      # M >= 4096 with N % 256 == 0: the LDS-tiled linear_big_kernel (forward and dX)
                      (4096 + 37, 256, 768), (5000, 1024, 256), (4608, 256, 1024), (4100, 16, 256)]:
        g = FlatGroup("t", DEV)
        g.add("w", (N, K))
        g.mark_weight("w")
        g.add("b", (N,))
        g.finalize()
        W = rs.randn(N, K).astype(np.float32) * 0.1
        b = rs.randn(N).astype(np.float32)
        g.view("w").copy_(t(W))
        g.view("b").copy_(t(b))
        g.repack()
        A = rs.randn(M, K).astype(np.float32)
        R = rs.randn(M, N).astype(np.float32)
        At, Rt, Y = t(A), t(R), torch.zeros(M, N, device=DEV)
        L.check(L.load().osrl_linear(At.data_ptr(), K, M, K, g.pf.data_ptr(), r16(N), 0, N, g.view("b").data_ptr(),
                                     Rt.data_ptr(), N, Y.data_ptr(), N, cur_stream()), "lin")
        ref = A.astype(np.float64) @ W.T.astype(np.float64) + b + R
        err = np.abs(Y.cpu().numpy() - ref).max()
        assert err < 3e-5 * max(1, np.abs(ref).max()), (M, K, N, "fwd", err)
        # dx = dY W  through the backward pack
        dY = rs.randn(M, N).astype(np.float32)
        dX = torch.zeros(M, K, device=DEV)
        dYt = t(dY)
        L.check(L.load().osrl_linear(dYt.data_ptr(), N, M, N, g.pb.data_ptr(), r16(K) + 16, 0, K, None, None, 0,
                                     dX.data_ptr(), K, cur_stream()), "lin dx")
        ref = dY.astype(np.float64) @ W.astype(np.float64)
        err = np.abs(dX.cpu().numpy() - ref).max()
        assert err < 3e-5 * max(1, np.abs(ref).max()), (M, K, N, "dx", err)


def test_linear_persistent_kernel_equals_the_tile_kernel_bit_for_bit():
    """The persistent 128 x 128-tile GEMM (M % 128 == 0, K % 256 == 0, N % 128 == 0, tiles spread evenly over the CUs)
    against float64 and, bit for bit, against linear_big_kernel (the same call with 16 more rows takes that path):
    residual / no residual, strided A / residual / Y, a column group of a wider pack (col0), several tiles per
    workgroup (drip of the previous tile during the next one) and K = 256 .. 1024 (1 .. 4 groups of 16 k-steps)."""
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import FlatGroup, cur_stream
    lib = L.load()
    rs = np.random.RandomState(5)
    r16 = lambda x: (x + 15) // 16 * 16  # noqa: E731
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for (M, K, N, Nw, col0, use_res, pad) in [(128 * ncu, 256, 128, 128, 0, True, 0),
                                             (128 * ncu // 2, 256, 256, 256, 0, False, 0),
                                             (64 * ncu, 1024, 256, 256, 0, True, 8),
                                             (64 * ncu, 768, 256, 768, 256, False, 4),
                                             (128 * ncu, 512, 384, 384, 0, True, 0),
                                             (128 * ncu * 3, 256, 128, 256, 128, True, 12)]:
        g = FlatGroup("t", DEV)
        g.add("w", (Nw, K))
        g.mark_weight("w")
        g.add("b", (Nw,))
        g.finalize()
        W, b = rs.randn(Nw, K).astype(np.float32) * 0.1, rs.randn(Nw).astype(np.float32)
        g.view("w").copy_(t(W))
        g.view("b").copy_(t(b))
        g.repack()
        Mx = M + 16
        lda, ldr, ldy = K + pad, N + pad, N + 2 * pad
        A = np.zeros((Mx, lda), np.float32)
        A[:, :K] = rs.randn(Mx, K)
        R = np.zeros((Mx, ldr), np.float32)
        R[:, :N] = rs.randn(Mx, N)
        At, Rt = t(A), t(R)
        bias = g.view("b")[col0:col0 + N].contiguous()
        outs = []
        for rows in (M, Mx):
            Y = torch.full((Mx, ldy), 7.0, device=DEV)
            L.check(lib.osrl_linear(At.data_ptr(), lda, rows, K, g.pf.data_ptr(), r16(Nw), col0, N, bias.data_ptr(),
                                    Rt.data_ptr() if use_res else None, ldr, Y.data_ptr(), ldy, cur_stream()), "lin")
            outs.append(Y.cpu().numpy())
        y_pers, y_tile = outs
        ref = A[:M, :K].astype(np.float64) @ W[col0:col0 + N].T.astype(np.float64) + b[col0:col0 + N]
        if use_res:
            ref = ref + R[:M, :N]
        assert np.abs(y_pers[:M, :N] - ref).max() < 3e-5 * max(1, np.abs(ref).max()), (M, K, N)
        assert np.array_equal(y_pers[:M, :N], y_tile[:M, :N]), (M, K, N)
        assert np.all(y_pers[:M, N:] == 7.0) and np.all(y_pers[M:] == 7.0), "writes outside the [M, N] block"


def test_layernorm_gelu_attention_kernels():
    from oracle.cdt_oracle import gelu, gelu_grad, layer_norm, layer_norm_bwd
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import cur_stream
    lib = L.load()
    rs = np.random.RandomState(1)
    for (M, E) in [(37, 16), (260, 128), (1000, 256), (301, 512)]:
        x, d = rs.randn(M, E), rs.randn(M, E) * 0.3
        gm, bt = 1 + 0.1 * rs.randn(E), 0.1 * rs.randn(E)
        f = lambda a: t(np.asarray(a, np.float32))  # noqa: E731
        xt, dt_, gt, btt = f(x), f(d), f(gm), f(bt)
        xo, y, st = torch.zeros(M, E, device=DEV), torch.zeros(M, E, device=DEV), torch.zeros(M, 2, device=DEV)
        L.check(lib.osrl_layernorm_fwd(xt.data_ptr(), dt_.data_ptr(), gt.data_ptr(), btt.data_ptr(), xo.data_ptr(),
                                       y.data_ptr(), st.data_ptr(), M, E, cur_stream()), "ln")
        xs = x.astype(np.float32).astype(np.float64) + d.astype(np.float32).astype(np.float64)
        yr, cache = layer_norm(xs, gm.astype(np.float32).astype(np.float64), bt.astype(np.float32).astype(np.float64))
        assert np.abs(y.cpu().numpy() - yr).max() < 2e-5 and np.abs(xo.cpu().numpy() - xs).max() < 1e-6
        dy, dres = rs.randn(M, E), rs.randn(M, E)
        dx = torch.zeros(M, E, device=DEV)
        nparts = 7
        ws, slab = torch.zeros(nparts, 2 * E, device=DEV), torch.zeros(4 * E + 8, device=DEV)
        dyt, drt = f(dy), f(dres)  # keep the tensors alive: data_ptr() of a temporary dangles
        L.check(lib.osrl_layernorm_bwd(dyt.data_ptr(), xo.data_ptr(), st.data_ptr(), gt.data_ptr(), drt.data_ptr(),
                                       dx.data_ptr(), ws.data_ptr(), nparts, M, E, slab.data_ptr(), 4, 4 + 2 * E,
                                       cur_stream()), "lnb")
        dxr, dgr, dbr = layer_norm_bwd(dy.astype(np.float32).astype(np.float64), cache,
                                       gm.astype(np.float32).astype(np.float64))
        dxr = dxr + dres.astype(np.float32)
        assert np.abs(dx.cpu().numpy() - dxr).max() < 5e-5, (M, E)
        sl = slab.cpu().numpy()
        assert np.abs(sl[4:4 + E] - dgr).max() < 1e-3 * max(1, np.abs(dgr).max())
        assert np.abs(sl[4 + 2 * E:4 + 3 * E] - dbr).max() < 1e-3 * max(1, np.abs(dbr).max())
    # gelu
    x = (rs.randn(4096) * 2).astype(np.float32)
    dy = rs.randn(4096).astype(np.float32)
    y, dx = torch.zeros(4096, device=DEV), torch.zeros(4096, device=DEV)
    xg, dyg = t(x), t(dy)
    L.check(lib.osrl_gelu_fwd(xg.data_ptr(), y.data_ptr(), 4096, cur_stream()), "g")
    L.check(lib.osrl_gelu_bwd(dyg.data_ptr(), xg.data_ptr(), dx.data_ptr(), 4096, cur_stream()), "gb")
    assert np.abs(y.cpu().numpy() - gelu(x.astype(np.float64))).max() < 2e-6
    assert np.abs(dx.cpu().numpy() - dy * gelu_grad(x.astype(np.float64))).max() < 5e-6
    # attention fwd/bwd vs numpy
    for (B, T, E, H) in [(3, 4, 16, 2), (2, 20, 256, 8), (5, 10, 128, 8)]:
        S, d = 4 * T, E // H
        qkv = rs.randn(B, S, 3 * E).astype(np.float32)
        mask = np.ones((B, T), np.float32)
        mask[0, T - 2:] = 0
        do = rs.randn(B, S, E).astype(np.float32)
        o, dqkv = torch.zeros(B, S, E, device=DEV), torch.zeros(B, S, 3 * E, device=DEV)
        qt, mt = t(qkv), t(mask)
        L.check(lib.osrl_attention_fwd(qt.data_ptr(), mt.data_ptr(), B, S, E, H, 4, 0, None, o.data_ptr(), cur_stream()),
                "a")
        dot = t(do)
        L.check(lib.osrl_attention_bwd(qt.data_ptr(), mt.data_ptr(), dot.data_ptr(), B, S, E, H, 4, 0, None,
                                       dqkv.data_ptr(), cur_stream()), "ab")
        q64 = qkv.astype(np.float64)
        q, k, v = (q64[..., i * E:(i + 1) * E].reshape(B, S, H, d).transpose(0, 2, 1, 3) for i in range(3))
        blocked = np.triu(np.ones((S, S), bool), 1)[None, None] | np.repeat(mask <= 0, 4, 1)[:, None, None, :]
        sc = np.where(blocked, -np.inf, q @ k.transpose(0, 1, 3, 2) / math.sqrt(d))
        P = np.exp(sc - sc.max(-1, keepdims=True))
        P /= P.sum(-1, keepdims=True)
        oref = (P @ v).transpose(0, 2, 1, 3).reshape(B, S, E)
        assert np.abs(o.cpu().numpy() - oref).max() < 2e-5, (B, T, E, H)
        dO = do.astype(np.float64).reshape(B, S, H, d).transpose(0, 2, 1, 3)
        dP = dO @ v.transpose(0, 1, 3, 2)
        dv = P.transpose(0, 1, 3, 2) @ dO
        dS = P * (dP - (dP * P).sum(-1, keepdims=True))
        dq, dk = dS @ k / math.sqrt(d), dS.transpose(0, 1, 3, 2) @ q / math.sqrt(d)
        ref = np.concatenate([x.transpose(0, 2, 1, 3).reshape(B, S, E) for x in (dq, dk, dv)], -1)
        assert np.abs(dqkv.cpu().numpy() - ref).max() < 5e-5 * max(1, np.abs(ref).max()), (B, T, E, H)


def test_layernorm_16_byte_kernels_equal_the_generic_ones():
    """E = 256 / 512 with 16-byte aligned tensors take the kernels whose lanes own four consecutive features; a 4-byte
    offset of one operand sends the same call to the generic kernels.  Element-wise results that involve no row
    reduction (x + delta * mask, and the positions the dropout site zeroes) are bit-equal; LayerNorm outputs agree to
    rounding (the lanes' partial sums meet in a different order)."""
    import ctypes as C
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import StepState, cur_stream
    lib = L.load()
    rs = np.random.RandomState(3)
    st = StepState(torch.device(DEV), ["x"])
    st.tick()
    for E in (256, 512):
        M, p, nparts = 333, 0.25, 64
        f = lambda a: t(np.asarray(a, np.float32))  # noqa: E731
        gm, bt = f(1 + 0.1 * rs.randn(E)), f(0.1 * rs.randn(E))
        x, dl, dy, dres = (rs.randn(M * E + 1).astype(np.float32) for _ in range(4))
        outs = []
        for off in (0, 1):  # 0: aligned operands; 1: x / dy start 4 bytes further (same values)
            xt, dyt = f(x[:M * E]), f(dy[:M * E])
            if off:
                xt = f(np.concatenate([[0.0], x[:M * E]]))[1:]
                dyt = f(np.concatenate([[0.0], dy[:M * E]]))[1:]
            dlt, drt = f(dl[:M * E]), f(dres[:M * E])
            xo, y, stt = torch.zeros(M * E, device=DEV), torch.zeros(M * E, device=DEV), torch.zeros(M, 2, device=DEV)
            d = L.DropoutT(p, 5, 77, st.ptr)
            L.check(lib.osrl_layernorm_fwd_drop(xt.data_ptr(), dlt.data_ptr(), C.byref(d), gm.data_ptr(), bt.data_ptr(),
                                                xo.data_ptr(), y.data_ptr(), stt.data_ptr(), M, E, cur_stream()), "lnf")
            dx, dxd = torch.zeros(M * E, device=DEV), torch.zeros(M * E, device=DEV)
            ws, slab = torch.zeros(nparts, 2 * E, device=DEV), torch.zeros(2 * E, device=DEV)
            L.check(lib.osrl_layernorm_bwd_drop(dyt.data_ptr(), xo.data_ptr(), stt.data_ptr(), gm.data_ptr(),
                                                drt.data_ptr(), dx.data_ptr(), dxd.data_ptr(), C.byref(d), ws.data_ptr(),
                                                nparts, M, E, slab.data_ptr(), 0, E, cur_stream()), "lnb")
            outs.append([v.cpu().numpy() for v in (xo, y, stt, dx, dxd, slab)])
        (xo0, y0, s0, dx0, dxd0, sl0), (xo1, y1, s1, dx1, dxd1, sl1) = outs
        assert np.array_equal(xo0, xo1)
        assert np.array_equal(dxd0 == 0, dxd1 == 0) and 0.15 < (dxd0 == 0).mean() < 0.35
        assert np.abs(y0 - y1).max() < 3e-6 and np.abs(s0 - s1).max() < 3e-6 * max(1, np.abs(s1).max())
        assert np.abs(dx0 - dx1).max() < 1e-5 and np.abs(dxd0 - dxd1).max() < 2e-5
        assert np.abs(sl0 - sl1).max() < 1e-4 * max(1, np.abs(sl1).max())


@pytest.mark.parametrize("use_rew,use_cost,prefix,E", [(1, 1, 0, 256), (0, 1, 1, 256), (1, 0, 0, 512)])
def test_embed_layernorm_16_byte_kernel_equals_the_generic_one(use_rew, use_cost, prefix, E):
    """osrl_cdt_embed_ln at E = 256 / 512 and >= 4096 token rows takes the kernel that stages the transposed state / action
    embedding weights in LDS and gives every lane four consecutive features; with `seq` 4 bytes off a 16-byte boundary the
    same call takes the generic kernel.  The embedded sequence is bit-equal, the LayerNorm output agrees to rounding."""
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import cur_stream
    lib = L.load()
    rs = np.random.RandomState(11 + E)
    B, T, od, ad = 72, 20, 11, 3
    R = 2 + use_rew + use_cost
    S = R * T + prefix
    f = lambda *shape: t((rs.randn(*shape) * 0.5).astype(np.float32))  # noqa: E731
    states, actions, returns, ctg, ec = f(B, T, od), f(B, T, ad), f(B, T), f(B, T), f(B)
    ts = t(rs.randint(0, 50, size=(B, T)).astype(np.int64))
    Ws, bs, Wa, ba, Wc, bc, Wr, br, Wp, bp = f(E, od), f(E), f(E, ad), f(E), f(E), f(E), f(E), f(E), f(E), f(E)
    te, g, b = f(50, E), f(E), f(E)
    outs = []
    for off in (0, 1):
        seq = torch.zeros(B * S * E + 4, device=DEV)[off:off + B * S * E]
        x0, stats, ctg_t = torch.zeros(B * S * E, device=DEV), torch.zeros(B * S, 2, device=DEV), torch.zeros(B * T, device=DEV)
        L.check(lib.osrl_cdt_embed_ln(states.data_ptr(), actions.data_ptr(), returns.data_ptr() if use_rew else None,
                                      ctg.data_ptr() if use_cost else None, ec.data_ptr() if prefix else None,
                                      ts.data_ptr(), Ws.data_ptr(), bs.data_ptr(), Wa.data_ptr(), ba.data_ptr(),
                                      Wc.data_ptr() if use_cost else None, bc.data_ptr() if use_cost else None,
                                      Wr.data_ptr() if use_rew else None, br.data_ptr() if use_rew else None,
                                      Wp.data_ptr() if prefix else None, bp.data_ptr() if prefix else None, te.data_ptr(),
                                      g.data_ptr(), b.data_ptr(), B, T, od, ad, E, 1, use_rew, use_cost, prefix,
                                      seq.data_ptr(), x0.data_ptr(), stats.data_ptr(),
                                      ctg_t.data_ptr() if use_cost else None, cur_stream()), "embed")
        outs.append([v.clone().cpu().numpy() for v in (seq, x0, stats, ctg_t)])
    (q0, x00, s0, c0), (q1, x01, s1, c1) = outs
    assert np.array_equal(q0, q1) and np.array_equal(c0, c1) and np.abs(q0).max() > 0.1
    assert np.abs(x00 - x01).max() < 5e-6 and np.abs(s0 - s1).max() < 5e-6 * max(1, np.abs(s1).max())


def test_dropout_kernels():
    """osrl_dropout: keep-rate, scale, determinism in (seed, step, site), odd sizes / unaligned pointers; attention
    probability dropout forward + backward against numpy with the exported mask."""
    import ctypes as C
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import StepState, cur_stream
    lib = L.load()
    st = StepState(torch.device(DEV), ["x"])
    st.tick()

    def mask(n, p, site, seed=5, off=0):
        x = torch.ones(n + off, device=DEV)[off:]
        y = torch.full((n + off,), -1.0, device=DEV)[off:]
        d = L.DropoutT(p, site, seed, st.ptr)
        L.check(lib.osrl_dropout(x.data_ptr(), y.data_ptr(), n, C.byref(d), cur_stream()), "drop")
        return y

    n = 1 << 20
    for p in (0.1, 0.5):
        m0 = mask(n, p, 3)
        vals = torch.unique(m0).cpu().numpy()
        assert np.allclose(vals, [0.0, 1.0 / (1.0 - p)], rtol=1e-6), vals
        keep = (m0 > 0).float().mean().item()
        assert abs(keep - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n), (p, keep)
        assert torch.equal(m0, mask(n, p, 3))                      # same (seed, step, site) -> same mask
        assert not torch.equal(m0, mask(n, p, 4))                  # other site
        assert not torch.equal(m0, mask(n, p, 3, seed=6))          # other seed
        assert torch.equal(m0[:1001], mask(1001, p, 3))            # prefix-stable, ragged tail
        assert torch.equal(m0[:1001], mask(1001, p, 3, off=1))     # unaligned pointers take the scalar path
    m0 = mask(n, 0.1, 3)
    st.tick()
    assert not torch.equal(m0, mask(n, 0.1, 3))                    # next train step -> fresh mask
    assert lib.osrl_dropout(m0.data_ptr(), m0.data_ptr(), n, C.byref(L.DropoutT(1.0, 0, 0, st.ptr)), cur_stream()) != 0

    rs = np.random.RandomState(2)
    for (B, T, E, H, p) in [(3, 4, 16, 2, 0.25), (2, 20, 256, 8, 0.1), (4, 24, 64, 2, 0.5)]:
        S, d = 4 * T, E // H
        qkv = rs.randn(B, S, 3 * E).astype(np.float32)
        mk = np.ones((B, T), np.float32)
        mk[0, T - 2:] = 0
        do = rs.randn(B, S, E).astype(np.float32)
        o, dqkv = torch.zeros(B, S, E, device=DEV), torch.zeros(B, S, 3 * E, device=DEV)
        qt, mt, dot = t(qkv), t(mk), t(do)
        dr = L.DropoutT(p, 7, 11, st.ptr)
        L.check(lib.osrl_attention_fwd(qt.data_ptr(), mt.data_ptr(), B, S, E, H, 4, 0, C.byref(dr), o.data_ptr(),
                                       cur_stream()), "a")
        L.check(lib.osrl_attention_bwd(qt.data_ptr(), mt.data_ptr(), dot.data_ptr(), B, S, E, H, 4, 0, C.byref(dr),
                                       dqkv.data_ptr(), cur_stream()), "ab")
        Sp = (S + 15) // 16 * 16  # the attention kernels' mask layout: [B*H, S, Sp], 4 consecutive keys per draw
        raw = torch.empty(B * H, S, Sp, device=DEV)
        ones = torch.ones_like(raw)
        L.check(lib.osrl_dropout(ones.data_ptr(), raw.data_ptr(), raw.numel(), C.byref(dr), cur_stream()), "m")
        Mk = raw.cpu().numpy()[:, :, :S].reshape(B, H, S, S).astype(np.float64)
        q64 = qkv.astype(np.float64)
        q, k, v = (q64[..., i * E:(i + 1) * E].reshape(B, S, H, d).transpose(0, 2, 1, 3) for i in range(3))
        blocked = np.triu(np.ones((S, S), bool), 1)[None, None] | np.repeat(mk <= 0, 4, 1)[:, None, None, :]
        sc = np.where(blocked, -np.inf, q @ k.transpose(0, 1, 3, 2) / math.sqrt(d))
        P = np.exp(sc - sc.max(-1, keepdims=True))
        P /= P.sum(-1, keepdims=True)
        Pd = P * Mk
        oref = (Pd @ v).transpose(0, 2, 1, 3).reshape(B, S, E)
        assert np.abs(o.cpu().numpy() - oref).max() < 3e-5, (B, T, E, H)
        dO = do.astype(np.float64).reshape(B, S, H, d).transpose(0, 2, 1, 3)
        dP = (dO @ v.transpose(0, 1, 3, 2)) * Mk
        dv = Pd.transpose(0, 1, 3, 2) @ dO
        dS = P * (dP - (dP * P).sum(-1, keepdims=True))
        dq, dk = dS @ k / math.sqrt(d), dS.transpose(0, 1, 3, 2) @ q / math.sqrt(d)
        ref = np.concatenate([x.transpose(0, 2, 1, 3).reshape(B, S, E) for x in (dq, dk, dv)], -1)
        assert np.abs(dqkv.cpu().numpy() - ref).max() < 5e-5 * max(1, np.abs(ref).max()), (B, T, E, H)


@pytest.mark.parametrize("S,E,H,rep,prefix,p", [
    (16, 128, 8, 4, 0, 0.1),    # 1 row block, head width 16: 2 waves per (sample, head)
    (40, 128, 8, 4, 0, 0.1),    # the reference's default CDT shape (seq_len 10, 128 / 8)
    (41, 256, 8, 4, 1, 0.1),    # cost-prefix token, ragged last row block
    (80, 256, 8, 4, 0, 0.1),    # C5: 5 row blocks on 3 waves
    (80, 256, 8, 4, 0, 0.0),
    (96, 256, 16, 4, 0, 0.2),   # 6 row blocks on 3 waves (the 8-block instantiation), head width 16
    (128, 256, 8, 2, 0, 0.1),   # 8 row blocks on 4 waves
    (81, 512, 8, 4, 1, 0.1),    # head width 64: the generic kernels
])
def test_attention_wave_splits_padding_and_dropout(S, E, H, rep, prefix, p):
    """Round 5's attention kernels (head widths 16 / 32: row block as a template parameter, 2 / 3 / 4 waves per (sample,
    head), base-2 softmax with additive masks) and the generic ones behind the same two entry points, against numpy:
    key padding at the tail AND at the front (query rows without any valid key give zero rows, as the oracle's masked
    softmax does), the cost-prefix token, probability dropout with the exported mask."""
    import ctypes as C
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import StepState, cur_stream
    lib = L.load()
    st = StepState(torch.device(DEV), ["x"])
    st.tick()
    rs = np.random.RandomState(S * 7 + E + H)
    B, d, T = 7, E // H, (S - prefix) // rep
    qkv = (0.7 * rs.randn(B, S, 3 * E)).astype(np.float32)
    mk = np.ones((B, T), np.float32)
    for b in range(B):
        pad = int(rs.randint(0, max(T - 1, 1)))
        if b % 3 == 0:
            mk[b, T - pad:] = 0
        elif b % 3 == 1:
            mk[b, :pad] = 0
    do = rs.randn(B, S, E).astype(np.float32)
    o, dqkv = torch.full((B, S, E), 7.0, device=DEV), torch.full((B, S, 3 * E), 7.0, device=DEV)
    qt, mt, dot = t(qkv), t(mk), t(do)
    dr = L.DropoutT(p, 9, 13, st.ptr)
    drp = C.byref(dr) if p > 0 else None
    L.check(lib.osrl_attention_fwd(qt.data_ptr(), mt.data_ptr(), B, S, E, H, rep, prefix, drp, o.data_ptr(), cur_stream()), "a")
    L.check(lib.osrl_attention_bwd(qt.data_ptr(), mt.data_ptr(), dot.data_ptr(), B, S, E, H, rep, prefix, drp,
                                   dqkv.data_ptr(), cur_stream()), "ab")
    Mk = np.ones((B, H, S, S))
    if p > 0:
        Sp = (S + 15) // 16 * 16
        raw, ones = torch.empty(B * H, S, Sp, device=DEV), torch.ones(B * H, S, Sp, device=DEV)
        L.check(lib.osrl_dropout(ones.data_ptr(), raw.data_ptr(), raw.numel(), drp, cur_stream()), "m")
        Mk = raw.cpu().numpy()[:, :, :S].reshape(B, H, S, S).astype(np.float64)
    key_ok = np.repeat(mk > 0, rep, 1)
    if prefix:
        key_ok = np.concatenate([key_ok[:, :1], key_ok], 1)  # the prefix token is masked like timestep 0 (cdt.py:216-218)
    q64 = qkv.astype(np.float64)
    q, k, v = (q64[..., i * E:(i + 1) * E].reshape(B, S, H, d).transpose(0, 2, 1, 3) for i in range(3))
    blocked = np.triu(np.ones((S, S), bool), 1)[None, None] | ~key_ok[:, None, None, :]
    sc = np.where(blocked, -np.inf, q @ k.transpose(0, 1, 3, 2) / math.sqrt(d))
    mx = sc.max(-1, keepdims=True)
    dead = ~np.isfinite(mx)
    P = np.exp(sc - np.where(dead, 0.0, mx))
    P = np.where(dead, 0.0, P / np.where(dead, 1.0, P.sum(-1, keepdims=True)))
    Pd = P * Mk
    oref = (Pd @ v).transpose(0, 2, 1, 3).reshape(B, S, E)
    assert np.abs(o.cpu().numpy() - oref).max() < 3e-5, "o"
    dO = do.astype(np.float64).reshape(B, S, H, d).transpose(0, 2, 1, 3)
    dP = (dO @ v.transpose(0, 1, 3, 2)) * Mk
    dv = Pd.transpose(0, 1, 3, 2) @ dO
    dS = P * (dP - (dP * P).sum(-1, keepdims=True))
    dq, dk = dS @ k / math.sqrt(d), dS.transpose(0, 1, 3, 2) @ q / math.sqrt(d)
    ref = np.concatenate([x.transpose(0, 2, 1, 3).reshape(B, S, E) for x in (dq, dk, dv)], -1)
    assert np.abs(dqkv.cpu().numpy() - ref).max() < 5e-5 * max(1, np.abs(ref).max()), "dqkv"


@pytest.mark.parametrize("S,E,H,rep,prefix,p", [(80, 256, 8, 4, 0, 0.1), (81, 128, 8, 4, 1, 0.3), (96, 256, 16, 4, 0, 0.2),
                                                 (36, 64, 2, 2, 0, 0.1), (128, 256, 8, 2, 0, 0.1)])
def test_attention_keep_handoff_is_bit_identical(S, E, H, rep, prefix, p):
    """osrl_attention_fwd_keep / _bwd_keep (round 6: the probability dropout's keep decisions written by the forward launch,
    one nibble per four keys, and read back by the backward launch instead of regenerated) == the plain pair bit for bit, at
    head widths 16 / 32, ragged S, the prefix token, 2 / 3 / 4 waves per pair; the bytes themselves == the exported Philox
    mask; a shape without the fast kernels reports 0 bytes and refuses a keep buffer."""
    import ctypes as C
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import StepState, cur_stream
    lib = L.load()
    st = StepState(torch.device(DEV), ["x"])
    st.tick()
    rs = np.random.RandomState(S + E)
    B, T = 5, (S - prefix) // rep
    qt = t((0.7 * rs.randn(B, S, 3 * E)).astype(np.float32))
    mk = np.ones((B, T), np.float32)
    mk[1, T - 3:] = 0
    mk[2, :2] = 0
    mt, dot = t(mk), t(rs.randn(B, S, E).astype(np.float32))
    dr = L.DropoutT(p, 9, 13, st.ptr)
    drp = C.byref(dr)
    nbytes = int(lib.osrl_attention_keep_bytes(B, S, E, H))
    Sp = (S + 15) // 16 * 16
    assert nbytes == B * H * (Sp // 16) * Sp * 4
    keep = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=DEV)
    o0, o1 = torch.zeros(B, S, E, device=DEV), torch.zeros(B, S, E, device=DEV)
    g0, g1 = torch.zeros(B, S, 3 * E, device=DEV), torch.zeros(B, S, 3 * E, device=DEV)
    L.check(lib.osrl_attention_fwd(qt.data_ptr(), mt.data_ptr(), B, S, E, H, rep, prefix, drp, o0.data_ptr(), cur_stream()), "f")
    L.check(lib.osrl_attention_bwd(qt.data_ptr(), mt.data_ptr(), dot.data_ptr(), B, S, E, H, rep, prefix, drp, g0.data_ptr(),
                                   cur_stream()), "b")
    L.check(lib.osrl_attention_fwd_keep(qt.data_ptr(), mt.data_ptr(), B, S, E, H, rep, prefix, drp, o1.data_ptr(),
                                        keep.data_ptr(), cur_stream()), "fk")
    L.check(lib.osrl_attention_bwd_keep(qt.data_ptr(), mt.data_ptr(), dot.data_ptr(), B, S, E, H, rep, prefix, drp,
                                        g1.data_ptr(), keep.data_ptr(), cur_stream()), "bk")
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(g0, g1)
    assert o0.abs().sum().item() > 0 and g0.abs().sum().item() > 0
    # the bytes against the exported mask: keep[bh][jb][row][quad] bit r <-> mask[bh, row, 16 jb + 4 quad + r] (lower triangle)
    raw, ones = torch.empty(B * H, S, Sp, device=DEV), torch.ones(B * H, S, Sp, device=DEV)
    L.check(lib.osrl_dropout(ones.data_ptr(), raw.data_ptr(), raw.numel(), drp, cur_stream()), "m")
    Mk = (raw.cpu().numpy() > 0)
    kb = keep.cpu().numpy().reshape(B * H, Sp // 16, Sp, 4)
    for jb in range(Sp // 16):
        rows = np.arange(16 * jb, S)  # row blocks ib >= jb hold key block jb
        for quad in range(4):
            for r in range(4):
                j = 16 * jb + 4 * quad + r
                want = Mk[:, rows, j]
                got = (kb[:, jb, rows, quad] >> r) & 1
                assert np.array_equal(got.astype(bool), want), (jb, quad, r)
    # head width 64: no fast kernels, no hand-off
    assert int(lib.osrl_attention_keep_bytes(B, 80, 512, 8)) == 0
    q2 = torch.zeros(B, 80, 3 * 512, device=DEV)
    o2 = torch.zeros(B, 80, 512, device=DEV)
    m2 = torch.ones(B, 20, device=DEV)
    assert lib.osrl_attention_fwd_keep(q2.data_ptr(), m2.data_ptr(), B, 80, 512, 8, 4, 0, drp, o2.data_ptr(), keep.data_ptr(),
                                       cur_stream()) != 0


def test_layernorm_param_reduce_and_counted_slab_sum():
    """osrl_layernorm_param_reduce (every LayerNorm's dgamma | dbeta in one launch) and osrl_reduce_slabs_counts (a split
    count per 1024-float chunk) return the bits of the per-call forms they replace in the CDT step."""
    import ctypes as C
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import cur_stream
    lib = L.load()
    g = torch.Generator(device="cpu").manual_seed(3)
    M, E, nparts, sites = 4096, 256, 128, 5
    slab_a, slab_b = torch.zeros(8192, device=DEV), torch.zeros(8192, device=DEV)
    ws = torch.zeros(sites, nparts, 2 * E, device=DEV)
    gam = (1 + 0.1 * torch.randn(E, generator=g)).to(DEV)
    offs = [(16 + 1000 * k, 16 + 1000 * k + 400) for k in range(sites)]
    for k in range(sites):
        dy, x = torch.randn(M, E, generator=g).to(DEV), torch.randn(M, E, generator=g).to(DEV)
        stats = torch.stack([x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], 1).contiguous()
        dx = torch.zeros(M, E, device=DEV)
        one = torch.zeros(nparts, 2 * E, device=DEV)
        L.check(lib.osrl_layernorm_bwd(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), gam.data_ptr(), None, dx.data_ptr(),
                                       one.data_ptr(), nparts, M, E, slab_a.data_ptr(), offs[k][0], offs[k][1], cur_stream()), "a")
        L.check(lib.osrl_layernorm_bwd(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), gam.data_ptr(), None, dx.data_ptr(),
                                       ws[k].data_ptr(), nparts, M, E, None, 0, 0, cur_stream()), "b")
    g_offs = (C.c_int64 * sites)(*[o[0] for o in offs])
    b_offs = (C.c_int64 * sites)(*[o[1] for o in offs])
    L.check(lib.osrl_layernorm_param_reduce(ws.data_ptr(), nparts * 2 * E, sites, nparts, E, slab_b.data_ptr(), g_offs, b_offs,
                                            cur_stream()), "many")
    assert torch.equal(slab_a, slab_b) and slab_a.abs().sum().item() > 0
    assert lib.osrl_layernorm_param_reduce(ws.data_ptr(), nparts * 2 * E, 17, nparts, E, slab_b.data_ptr(), g_offs, b_offs,
                                           cur_stream()) != 0
    # counted slab sum: chunk c holds gradients in its first counts[c] slabs, zeros behind
    n, S = 10 * 1024 + 512, 8
    counts = torch.tensor([1, 8, 3, 2, 8, 1, 5, 7, 4, 2, 6], dtype=torch.uint8)
    slabs = torch.randn(S, n, generator=g)
    for c, k in enumerate(counts.tolist()):
        slabs[k:, 1024 * c:1024 * (c + 1)] = 0
    s1, s2 = slabs.clone().to(DEV), slabs.clone().to(DEV)
    cd = counts.to(DEV)
    L.check(lib.osrl_reduce_slabs(s1.data_ptr(), s1.data_ptr(), S, n, n, cur_stream()), "r")
    L.check(lib.osrl_reduce_slabs_counts(s2.data_ptr(), s2.data_ptr(), cd.data_ptr(), n, n, cur_stream()), "rc")
    assert torch.equal(s1[0], s2[0])
    ref = slabs.double().sum(0)
    assert (s2[0].cpu().double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("case", ["cdt_c5_slice", "cdt_v_prefix", "cdt_small"])
def test_counted_slab_sum_equals_full_slab_sum_on_a_real_step(case):
    """``osrl_reduce_slabs_counts`` sums only each range's OWN split count per 1024-float chunk -- correct only
    while every slab row beyond a range's count is zero.  On the complete slabs of a REAL CDT step (token-matrix plan,
    per-sample plan, the prefix plan, LayerNorm / timestep rows written straight into slab 0) the counted sum must equal the
    sum over ALL slab rows bit for bit; and no slab row beyond a chunk's count may hold a non-zero (a writer missing from
    ``DwPlan.split_ranges()`` would show up here as dropped gradient)."""
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import cur_stream
    c = C5_SLICE if case == "cdt_c5_slice" else CDT_CASES[case]
    m, tr, lg = build_cdt_gpu(c, seed=1234)
    b = {k: t(v) for k, v in make_cdt_batch(c).items()}
    eng = m.engine(c.B, tr.cfg)
    seen = []

    def probe(slabs, n_splits, counts):
        lib = L.load()
        assert counts is not None, "the counted slab sum must be the shipped path"
        S, n = slabs.shape
        a, bb = slabs.clone(), slabs.clone()
        L.check(lib.osrl_reduce_slabs(a.data_ptr(), a.data_ptr(), S, n, n, cur_stream()), "r")
        L.check(lib.osrl_reduce_slabs_counts(bb.data_ptr(), bb.data_ptr(), counts.data_ptr(), n, n, cur_stream()), "rc")
        torch.cuda.synchronize()
        cnt = counts.cpu().numpy().astype(np.int64)
        rows = np.arange(S)[:, None] >= np.repeat(cnt, 1024)[None, :n]  # [S, n]: True = beyond the chunk's count
        stray = int((slabs.cpu().numpy()[rows] != 0).sum())
        seen.append((bool(torch.equal(a[0], bb[0])), stray, S, int(cnt.max()), float(a[0].abs().sum())))

    eng._slab_probe = probe
    for _ in range(2):
        tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                          b["episode_cost"], b["costs"])
    eng._slab_probe = None
    assert len(seen) == 2
    for eq, stray, S, cmax, mass in seen:
        assert eq, "counted slab sum differs from the sum over all slab rows"
        assert stray == 0, f"{stray} non-zero gradient words beyond their chunk's split count"
        assert mass > 0 and cmax <= S


def build_cdt_gpu(c, **kw):
    from osrl_amd.algorithms import CDT, CDTTrainer
    from osrl_amd.common.logger import DummyLogger
    m = CDT(c.od, c.ad, 1.0, seq_len=c.T, episode_len=c.episode_len, embedding_dim=c.E, num_layers=c.layers,
            num_heads=c.heads, attention_dropout=c.dropout, residual_dropout=c.dropout, embedding_dropout=c.dropout,
            time_emb=c.time_emb, use_rew=c.use_rew, use_cost=c.use_cost, cost_transform=c.cost_transform,
            add_cost_feat=c.add_cost_feat, mul_cost_feat=c.mul_cost_feat, cat_cost_feat=c.cat_cost_feat,
            action_head_layers=c.head_layers, cost_prefix=c.cost_prefix, stochastic=c.stochastic,
            init_temperature=0.1, target_entropy=-c.ad, device=DEV)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_cdt_params(c).items()})
    lg = DummyLogger()
    args = dict(stats_mode="sync", use_graph=False)
    args.update(kw)
    tr = CDTTrainer(m, None, lg, learning_rate=c.lr, weight_decay=c.wd, clip_grad=c.clip, lr_warmup_steps=c.warmup,
                    reward_scale=0.1, loss_cost_weight=c.cost_w, loss_state_weight=c.state_w, device=DEV, **args)
    return m, tr, lg


@pytest.mark.parametrize("name", [n for n, c in CDT_CASES.items() if c.dropout == 0])
def test_cdt_train_step_matches_golden_and_oracle(name):
    from test_oracle_cdt_golden import build_cdt_oracle
    c = CDT_CASES[name]
    g = load_golden(name)
    keys = [str(k) for k in g["stat_keys"]]
    m, tr, lg = build_cdt_gpu(c)
    o = build_cdt_oracle(c)
    b = {k: t(v) for k, v in make_cdt_batch(c).items()}
    bn = make_cdt_batch(c)
    for s in range(c.steps):
        tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                          b["episode_cost"], b["costs"])
        ost = o.train_one_step(bn["states"], bn["actions"], bn["returns"], bn["costs_return"], bn["time_steps"],
                               bn["mask"], bn["episode_cost"], bn["costs"])
        ref = dict(zip(keys, g["stats"][s]))
        tol = 1e-5 if s == 0 else 1e-4
        for k in keys:
            got = lg.last("train/" + k)
            for nm, r in (("golden", ref[k]), ("oracle", ost[k])):
                assert abs(got - r) <= tol * max(1.0, abs(r)), f"{name} step {s} {k}: gpu {got} vs {nm} {r}"
        if f"s{s + 1}/log_temperature" in g:
            assert abs(m.log_temperature.item() - float(g[f"s{s + 1}/log_temperature"])) < 1e-6
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        for k, v in sd.items():
            if f"p{s + 1}/{k}" in g:
                d = np.abs(v - g[f"p{s + 1}/{k}"]).max()
                assert d <= 2e-5, f"{name} step {s + 1} param {k}: max diff {d:.3e}"
            elif f"p{s + 1}/smp/{k}" in g:
                d = np.abs(v.reshape(-1)[::97] - g[f"p{s + 1}/smp/{k}"]).max()
                assert d <= 2e-5, f"{name} step {s + 1} param sample {k}: {d:.3e}"
    ap, cp, sp = m(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"],
                   ~b["mask"].to(torch.bool), b["episode_cost"])
    a = (ap.mean if c.stochastic else ap).cpu().numpy()
    assert np.abs(a - g["act"]).max() <= 1e-4


# BASELINE.json C5 architecture (T=20 -> 80 tokens, E=256, 8 heads, 3 layers, dropout 0.1) on a 64-sample slice of
# the batch: 5120 token rows reach linear_big_kernel (M >= 4096) and the bench-size attention / dW tiles.
C5_SLICE = CDTCase("cdt_c5_slice", od=11, ad=3, B=64, T=20, E=256, heads=8, layers=3, episode_len=1000, steps=2,
                   warmup=500, dropout=0.1, seed=5)


@pytest.mark.parametrize("case,use_graph", [("cdt_drop", False), ("cdt_drop", True), ("cdt_c5_slice", False),
                                            ("cdt_v_prefix", False), ("cdt_v_prefix", True)])
def test_cdt_dropout_train_step_matches_oracle(case, use_graph):
    """Dropout 0.1 at every site (the train-config default): the GPU step draws Philox masks; the oracle (pinned
    against the reference with injected masks, tests/golden/cdt_drop.npz) replays the same masks, exported from
    the generator after each step.  Also: eval-mode inference applies no dropout."""
    from test_oracle_cdt_golden import build_cdt_oracle
    c = C5_SLICE if case == "cdt_c5_slice" else CDT_CASES[case]
    m, tr, lg = build_cdt_gpu(c, use_graph=use_graph, seed=1234)
    o = build_cdt_oracle(c)
    bn = make_cdt_batch(c)
    b = {k: t(v) for k, v in bn.items()}
    prev = None
    for s in range(c.steps):
        tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                          b["episode_cost"], b["costs"])
        masks = {k: v.cpu().numpy() for k, v in m.engine(c.B).dropout_masks().items()}
        assert set(masks) == {"emb"} | {f"{k}{l}" for l in range(c.layers) for k in ("attn", "res1_", "res2_")}
        if prev is not None:
            assert any((masks[k] != prev[k]).any() for k in masks), "dropout masks must change from step to step"
        prev = masks
        ost = o.train_one_step(bn["states"], bn["actions"], bn["returns"], bn["costs_return"], bn["time_steps"],
                               bn["mask"], bn["episode_cost"], bn["costs"], drop=masks)
        for k, r in ost.items():
            got = lg.last("train/" + k)
            assert abs(got - r) <= 1e-4 * max(1.0, abs(r)), f"step {s} {k}: gpu {got} vs oracle {r}"
        assert abs(m.log_temperature.item() - o.log_temperature) < 1e-6
        for k, v in m.state_dict().items():
            if v.dtype != torch.bool:
                d = np.abs(v.cpu().numpy() - o.p[k]).max()
                assert d <= 2e-5, f"step {s + 1} param {k}: max diff {d:.3e}"
    m.eval()
    ap, _, _ = m(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"],
                 ~b["mask"].to(torch.bool), b["episode_cost"])
    ref = o.act_mean(bn["states"], bn["actions"], bn["returns"], bn["costs_return"], bn["time_steps"], bn["mask"],
                     bn["episode_cost"])
    assert np.abs(ap.mean.cpu().numpy() - ref).max() <= 1e-4
    m.train()
    a1 = m(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], ~b["mask"].to(torch.bool),
           b["episode_cost"])[0].mean
    assert (a1 - ap.mean).abs().max() > 1e-6, "a model in train() mode applies dropout in forward"


C5_FULL = CDTCase("cdt_c5_full", od=11, ad=3, B=1024, T=20, E=256, heads=8, layers=3, episode_len=1000, steps=2,
                  warmup=500, dropout=0.1, seed=6)


def chunked_oracle_check(o, bn, masks, got, grp, *, ad, od, clip, cost_w, state_w, temp, label, CH=64):
    """Step-1 statistics and the full-batch GRADIENT of a CDT step against the fp64 oracle ``o`` (which must hold the
    parameters the step started from), on the batch ``bn`` (numpy) with the device's own dropout keep-multipliers ``masks``
    replayed.  The forward is per-sample independent and, with the global normalisers fixed, the loss is a sum over samples:
    the host evaluates forward and backward in ``CH``-sample chunks (``grads_only``, ``norm``), sums the chunks' gradients,
    clips by their global norm (cdt.py:398-399) and compares with the device's Adam first moments / (1 - beta1) per tensor.
    ``got``: the device's logged statistics by short key.  Returns (worst diff / scale, clip coefficient, tensors compared)."""
    import math
    B, T = bn["mask"].shape
    f8 = lambda a: np.asarray(a, np.float64)  # noqa: E731
    acc = dict(ll=0.0, ent=0.0, nv=0.0, cl=0.0, hit=0.0, msum=0.0, sl=0.0)
    norm = dict(nv=max(int((bn["mask"] > 0).sum()), 1) * ad, bt=B * T, state_n=B * (T - 1) * od, act_n=B * T * ad)
    gsum = {}
    for i in range(0, B, CH):
        sl = slice(i, i + CH)
        drop = {k: v[sl].cpu().numpy().astype(np.float64) for k, v in masks.items()}
        st_, ac_, mk_ = f8(bn["states"][sl]), f8(bn["actions"][sl]), f8(bn["mask"][sl])
        gch = o.train_one_step(st_, ac_, f8(bn["returns"][sl]), f8(bn["costs_return"][sl]), bn["time_steps"][sl], mk_,
                               f8(bn["episode_cost"][sl]), bn["costs"][sl], drop=drop, norm=norm, grads_only=True)
        for k, v in gch.items():
            gsum[k] = gsum.get(k, 0.0) + np.asarray(v, np.float64)
        res, _ = o.forward(st_, ac_, f8(bn["returns"][sl]), f8(bn["costs_return"][sl]), bn["time_steps"][sl], mk_, drop)
        valid = (mk_ > 0)[..., None]
        zz = (ac_ - res["mu"]) / np.exp(res["ls"])
        acc["ll"] += ((-0.5 * zz * zz - res["ls"] - 0.5 * math.log(2 * math.pi)) * valid).sum()
        acc["ent"] += ((0.5 + 0.5 * math.log(2 * math.pi) + res["ls"]) * valid).sum()
        acc["nv"] += valid.sum() * ad
        ci = bn["costs"][sl].astype(np.int64)
        lp = res["cost_logp"]
        acc["cl"] += (-(np.take_along_axis(lp, ci[..., None], -1)[..., 0]) * mk_).sum()
        acc["hit"] += ((lp.argmax(-1) == ci) * mk_).sum()
        acc["msum"] += mk_.sum()
        diff = res["state_pred"][:, :-1] - st_[:, 1:]
        acc["sl"] += ((diff ** 2) * mk_[:, :-1, None]).sum()
    want = dict(nll=-acc["ll"] / acc["nv"], ent=acc["ent"] / acc["nv"], cost_loss=acc["cl"] / (B * T),
                cost_acc=acc["hit"] / acc["msum"], state_loss=acc["sl"] / (B * (T - 1) * od))
    want["act_loss"] = want["nll"] - temp * want["ent"]
    want["all_loss"] = want["act_loss"] + cost_w * want["cost_loss"] + state_w * want["state_loss"]
    for k, r in want.items():
        assert abs(got[k] - r) <= 1e-4 * max(1.0, abs(r)), f"{label} {k}: gpu {got[k]} vs fp64 oracle forward {r}"
    # gradients: clip by the global norm (cdt.py:398-399), compare with Adam's first moments after this ONE step
    tot = math.sqrt(sum(float((v ** 2).sum()) for v in gsum.values()))
    coef = min(1.0, clip / (tot + 1e-6)) if clip is not None else 1.0
    worst, n_cmp = 0.0, 0
    for k, gv in gsum.items():
        gk = "cdt." + k
        if gk not in grp.layout or gk in grp.aliases:
            continue
        want_m = (1.0 - 0.9) * coef * gv
        got_m = grp._view(grp.m, gk).cpu().numpy().astype(np.float64)
        scale = max(np.abs(want_m).max(), 1e-12)
        d = np.abs(got_m - want_m.reshape(got_m.shape)).max()
        worst = max(worst, d / scale)
        n_cmp += 1
        assert d <= 2e-5 * scale + 1e-9, f"{label}, Adam first moment {k}: max diff {d:.3e} vs scale {scale:.3e}"
    assert n_cmp == len(gsum), (n_cmp, len(gsum), sorted(set(gsum) - {k[4:] for k in grp.layout}))
    return worst, coef, n_cmp


def test_cdt_c5_full_batch_forward_stats_and_graph():
    """BASELINE.json C5 at its full batch (B = 1024 -> 81920 token rows): the only run of linear_big_kernel's
    640-workgroup grids, the 81920-row dW launch and the 8192 attention workgroups under test.
    (a) step-1 logged statistics vs a forward-only fp64 pass of the pinned oracle with the GPU's own dropout masks
    replayed (the forward is per-sample independent, so the host evaluates it in 64-sample chunks);
    (b) finite statistics; (c) captured graph == eager launches after 2 steps;
    (d) (round 3) the full-batch GRADIENT: with the global normalisers fixed the loss is a sum over samples, so the fp64
    oracle's backward runs in the same 64-sample chunks (``grads_only``, ``norm``) and the chunks' gradients add up to
    the batch's; clipped by their global norm they must equal the GPU's Adam first moments / (1 - beta1) per tensor --
    this is the parity run of the 81920-row dW launch, the 8192-workgroup attention backward and the clip."""
    import math
    from test_oracle_cdt_golden import build_cdt_oracle
    c = C5_FULL
    bn = make_cdt_batch(c)
    b = {k: t(v) for k, v in bn.items()}
    args = (b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"], b["episode_cost"],
            b["costs"])
    m, tr, lg = build_cdt_gpu(c, use_graph=False, seed=77)
    o = build_cdt_oracle(c, np.float64)
    tr.train_one_step(*args)
    got = {k: lg.last("train/" + k) for k in ("nll", "ent", "cost_loss", "cost_acc", "state_loss", "act_loss", "all_loss")}
    assert all(np.isfinite(v) for v in got.values()), got
    masks = m.engine(c.B).dropout_masks()
    worst, coef, n_cmp = chunked_oracle_check(o, bn, masks, got, m.groups["cdt"], ad=c.ad, od=c.od, clip=c.clip, cost_w=c.cost_w,
                                              state_w=c.state_w, temp=0.1, label="C5 full batch")
    print(f"C5 full batch: {n_cmp} tensors, worst first-moment diff / scale {worst:.2e}, clip coefficient {coef:.4f}")
    del masks, m, tr
    torch.cuda.empty_cache()
    res = []
    for use_graph in (False, True):
        m, tr, lg = build_cdt_gpu(c, stats_mode="none", use_graph=use_graph, seed=77)
        for s in range(2):
            tr.train_one_step(*args)
        torch.cuda.synchronize()
        assert (m._engine.graph is not None) == use_graph
        st = m._engine.st.read_stats()
        assert all(np.isfinite(v) for v in st.values()), st
        res.append({k: v.clone() for k, v in m.state_dict().items()})
        del m, tr
        torch.cuda.empty_cache()
    for k in res[0]:
        if res[0][k].dtype == torch.bool:
            assert torch.equal(res[0][k], res[1][k])
        else:  # the timestep-embedding scatter uses fp32 atomics: order-dependent rounding only
            assert (res[0][k] - res[1][k]).abs().max() < 1e-6, k


@pytest.mark.parametrize("case", ["cdt_drop", "cdt_v_prefix"])
def test_cdt_dropout_folded_into_layernorm_equals_separate_passes(case, monkeypatch):
    """osrl_layernorm_fwd_drop / osrl_layernorm_bwd_drop (the residual branches' nn.Dropout applied by the LayerNorm
    launches) == separate osrl_dropout passes over 3 train steps (parameters and moments; same products and sums)."""
    import osrl_amd.engine.cdt as ce
    c = CDT_CASES[case]
    b = {k: t(v) for k, v in make_cdt_batch(c).items()}
    res = []
    for fuse in (True, False):
        monkeypatch.setattr(ce, "FUSE_DROP", fuse)
        m, tr, lg = build_cdt_gpu(c, use_graph=False, seed=99)
        for s_ in range(3):
            tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                              b["episode_cost"], b["costs"])
        torch.cuda.synchronize()
        g = m.groups["cdt"]
        res.append((g.p.clone(), g.m.clone(), g.v.clone()))
    # (not torch.equal: the timestep-embedding scatter accumulates with fp32 atomics, whose order -- and through the clip
    # norm every update's last bits -- varies from run to run; the same bound as graph == eager)
    for x, y, name in zip(res[0], res[1], ("p", "m", "v")):
        assert (x - y).abs().max() < 1e-6, name


def test_cdt_graph_replay_matches_eager():
    c = CDT_CASES["cdt_small"]
    res = []
    for use_graph in (False, True):
        m, tr, lg = build_cdt_gpu(c, stats_mode="none", use_graph=use_graph)
        b = {k: t(v) for k, v in make_cdt_batch(c).items()}
        for s in range(3):
            tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                              b["episode_cost"], b["costs"])
        torch.cuda.synchronize()
        res.append({k: v.clone() for k, v in m.state_dict().items()})
    for k in res[0]:
        if res[0][k].dtype == torch.bool:
            assert torch.equal(res[0][k], res[1][k])
        else:  # the timestep-embedding scatter uses fp32 atomics: order-dependent rounding only
            assert (res[0][k] - res[1][k]).abs().max() < 1e-6, k


def test_cdt_data_parallel_world1_matches_single(nccl_world1):
    """CDT DP wiring (global count normalisers, flat-gradient all-reduce before the clip, entropy/stat reductions,
    hipGraph capture incl. RCCL) as a 1-rank NCCL job == the plain single-GPU step."""
    import os
    import torch.distributed as dist
    from osrl_amd.engine.dist import DataParallel
    c = CDT_CASES["cdt_small"]
    assert nccl_world1.is_initialized()  # the session's one 1-rank NCCL group (conftest.py)
    try:
        res = []
        for mode in ("single", "dp_eager", "dp_graph"):
            m, tr, lg = build_cdt_gpu(c, use_graph=(mode == "dp_graph"))
            if mode != "single":
                m.engine(c.B, tr.cfg, dist=DataParallel())
            b = {k: t(v) for k, v in make_cdt_batch(c).items()}
            for s in range(3):
                tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"],
                                  b["mask"], b["episode_cost"], b["costs"])
            torch.cuda.synchronize()
            res.append(({k: v.clone() for k, v in m.state_dict().items()}, {k: [float(x) for x in v] for k, v in lg.data.items()}))
        for other in res[1:]:
            for k in res[0][0]:
                if res[0][0][k].dtype != torch.bool:
                    assert (res[0][0][k] - other[0][k]).abs().max() < 1e-6, k
            for k in res[0][1]:
                assert np.allclose(res[0][1][k], other[1][k], rtol=1e-5, atol=1e-6), k
    finally:
        pass


def test_cdt_checkpoint_resume_is_bit_identical(tmp_path):
    """CDT with dropout 0.1: the masks, the LR warm-up and the temperature optimizer all hang off the step count
    the checkpoint carries (osrl_amd/common/checkpoint.py)."""
    from osrl_amd.common.checkpoint import load_checkpoint, save_checkpoint
    c = CDT_CASES["cdt_drop"]
    b = {k: t(v) for k, v in make_cdt_batch(c).items()}

    def run(tr, n):
        for _ in range(n):
            tr.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                              b["episode_cost"], b["costs"])

    m_a, tr_a, _ = build_cdt_gpu(c)
    run(tr_a, 5)
    m_b, tr_b, _ = build_cdt_gpu(c)
    run(tr_b, 3)
    path = str(tmp_path / "cdt.pt")
    save_checkpoint(m_b, path)
    m_c, tr_c, _ = build_cdt_gpu(c)
    load_checkpoint(m_c, path)
    run(tr_c, 2)
    torch.cuda.synchronize()
    for k, v in m_a.state_dict().items():
        assert torch.equal(v, m_c.state_dict()[k]), k
    g_a, g_c = m_a.groups["cdt"], m_c.groups["cdt"]
    assert torch.equal(g_a.m, g_c.m) and torch.equal(g_a.v, g_c.v)
    assert torch.equal(m_a.log_temperature, m_c.log_temperature)
    assert torch.equal(m_a._engine.temp_mv, m_c._engine.temp_mv)


def test_linear_and_attention_random_shapes():
    """Seeded random shapes through osrl_linear (both packs, incl. the LDS-tiled big path and ragged M / K / N) and
    osrl_attention_fwd/bwd (head dims 4..64, S up to 96, tail padding in several rows)."""
    from osrl_amd import _lib as L
    from osrl_amd.engine.core import FlatGroup, cur_stream
    lib = L.load()
    rs = np.random.RandomState(77)
    r16 = lambda x: (x + 15) // 16 * 16  # noqa: E731
    for _ in range(14):
        M = int(rs.choice([1, 5, 16, 63, 257, 1000, 4096, 4500, 6001]))
        K = int(rs.choice([1, 3, 16, 30, 64, 200, 256, 448]))
        N = int(rs.choice([1, 2, 6, 16, 40, 256, 300, 512, 768]))
        g = FlatGroup("t", DEV)
        g.add("w", (N, K))
        g.mark_weight("w")
        g.add("b", (N,))
        g.finalize()
        W, b = rs.randn(N, K).astype(np.float32) * 0.1, rs.randn(N).astype(np.float32)
        g.view("w").copy_(t(W))
        g.view("b").copy_(t(b))
        g.repack()
        A, R = rs.randn(M, K).astype(np.float32), rs.randn(M, N).astype(np.float32)
        At, Rt, Y = t(A), t(R), torch.zeros(M, N, device=DEV)
        L.check(lib.osrl_linear(At.data_ptr(), K, M, K, g.pf.data_ptr(), r16(N), 0, N, g.view("b").data_ptr(),
                                Rt.data_ptr(), N, Y.data_ptr(), N, cur_stream()), "lin")
        ref = A.astype(np.float64) @ W.T.astype(np.float64) + b + R
        assert np.abs(Y.cpu().numpy() - ref).max() < 3e-5 * max(1, np.abs(ref).max()), (M, K, N, "fwd")
        dY = rs.randn(M, N).astype(np.float32)
        dX, dYt = torch.zeros(M, K, device=DEV), t(dY)
        L.check(lib.osrl_linear(dYt.data_ptr(), N, M, N, g.pb.data_ptr(), r16(K) + 16, 0, K, None, None, 0,
                                dX.data_ptr(), K, cur_stream()), "lin dx")
        ref = dY.astype(np.float64) @ W.astype(np.float64)
        assert np.abs(dX.cpu().numpy() - ref).max() < 3e-5 * max(1, np.abs(ref).max()), (M, K, N, "dx")
    for _ in range(8):
        H = int(rs.choice([1, 2, 4, 8]))
        d = int(rs.choice([4, 8, 16, 24, 32, 64]))
        T = int(rs.choice([1, 2, 5, 12, 20, 24]))
        B = int(rs.randint(1, 6))
        E, S = H * d, 4 * T
        qkv = rs.randn(B, S, 3 * E).astype(np.float32)
        mask = np.ones((B, T), np.float32)
        for bb in range(B):
            mask[bb, T - int(rs.randint(0, T)):] = 0  # 0 .. T-1 padded steps at the tail
        do = rs.randn(B, S, E).astype(np.float32)
        o, dqkv = torch.zeros(B, S, E, device=DEV), torch.zeros(B, S, 3 * E, device=DEV)
        qt, mt, dot = t(qkv), t(mask), t(do)
        L.check(lib.osrl_attention_fwd(qt.data_ptr(), mt.data_ptr(), B, S, E, H, 4, 0, None, o.data_ptr(), cur_stream()), "a")
        L.check(lib.osrl_attention_bwd(qt.data_ptr(), mt.data_ptr(), dot.data_ptr(), B, S, E, H, 4, 0, None,
                                       dqkv.data_ptr(), cur_stream()), "ab")
        q64 = qkv.astype(np.float64)
        q, k, v = (q64[..., i * E:(i + 1) * E].reshape(B, S, H, d).transpose(0, 2, 1, 3) for i in range(3))
        blocked = np.triu(np.ones((S, S), bool), 1)[None, None] | np.repeat(mask <= 0, 4, 1)[:, None, None, :]
        sc = np.where(blocked, -np.inf, q @ k.transpose(0, 1, 3, 2) / math.sqrt(d))
        P = np.exp(sc - sc.max(-1, keepdims=True))
        P /= P.sum(-1, keepdims=True)
        oref = (P @ v).transpose(0, 2, 1, 3).reshape(B, S, E)
        assert np.abs(o.cpu().numpy() - oref).max() < 2e-5, (B, T, E, H, "o")
        dO = do.astype(np.float64).reshape(B, S, H, d).transpose(0, 2, 1, 3)
        dP = dO @ v.transpose(0, 1, 3, 2)
        dv = P.transpose(0, 1, 3, 2) @ dO
        dS = P * (dP - (dP * P).sum(-1, keepdims=True))
        dq, dk = dS @ k / math.sqrt(d), dS.transpose(0, 1, 3, 2) @ q / math.sqrt(d)
        ref = np.concatenate([x.transpose(0, 2, 1, 3).reshape(B, S, E) for x in (dq, dk, dv)], -1)
        assert np.abs(dqkv.cpu().numpy() - ref).max() < 5e-5 * max(1, np.abs(ref).max()), (B, T, E, H, "dqkv")
