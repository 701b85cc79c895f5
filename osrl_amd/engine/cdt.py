"""The Constrained Decision Transformer train step as a static launch plan on MI355X.

Follows ``CDTTrainer.train_one_step`` (osrl/algorithms/cdt.py:343-418) around ``CDT.forward``
(:166-265) and ``TransformerBlock.forward`` (osrl/common/net.py:422-441): embeddings -> emb LayerNorm ->
num_layers x [LN, QKV, causal+padding attention, out-proj, residual, LN, Linear-GELU-Linear, residual]
-> out LayerNorm -> heads -> losses -> backward of all of it -> clip_grad_norm_ -> AdamW (warm-up LR)
-> temperature Adam.  Projections and their dX / dW GEMMs run on the packed-weight fp32-MFMA kernels
(csrc/mlp.hip: osrl_linear, osrl_mlp_backward_dw); everything else is csrc/cdt.hip.
Dropout (embedding cdt.py:222, attention probabilities net.py:406-409, residual net.py:414,439; train-config
default 0.1, examples/configs/cdt_configs.py:28-30) uses stateless Philox masks keyed by (seed, step, site, element):
the backward kernels regenerate them, nothing is stored; with p = 0 no extra kernel is launched.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional

import torch

from .. import _lib as L
from . import plan as _plan
from .core import DwPlan, FlatGroup, StepState, capture_step, cur_stream, load_into, check_plans_current

LN_BATCH = _plan.knob("OSRL_CDT_LN_BATCH", "1", "CDT: the LayerNorm parameter reductions of a step in one launch") == "1"
ATTN_KEEP = _plan.knob("OSRL_CDT_ATTN_KEEP", "1", "CDT: attention dropout decisions handed from forward to backward") == "1"
SLAB_COUNTS = _plan.knob("OSRL_CDT_SLAB_COUNTS", "1", "CDT: gradient slabs summed per range by its own split count") == "1"
FUSE_DROP = _plan.knob("OSRL_CDT_FUSE_DROP", "1", "residual-branch dropout inside the LayerNorm launches") == "1"  # residual-branch dropout inside the LayerNorm launches
STAT_KEYS = ["nll", "ent", "ent_reg", "all_loss", "act_loss", "cost_loss", "cost_acc", "state_loss", "train_lr"]


def _r16(x: int) -> int:
    return (x + 15) // 16 * 16


class CDTEngine:
    def __init__(self, model, batch_size: int, trainer_cfg: dict, dist=None, inference: bool = False):
        m = self.model = model
        self.cfg = trainer_cfg
        self.dist = dist
        B, T, E, H, NL = int(batch_size), m.seq_len, m.embedding_dim, m.num_heads, m.num_layers
        self.B, self.T, self.E, self.H, self.NL = B, T, E, H, NL
        od, ad = m.state_dim, m.action_dim
        dev = torch.device(m.device)
        self.dev = dev
        # token layout (cdt.py:96-112,185-218): R tokens per timestep = [return] [cost] state action, optional prefix token
        R = self.R = m.seq_repeat
        P = self.P = 1 if m.cost_prefix else 0
        self.slot_rew = 0 if m.use_rew else None
        self.slot_cost = (1 if m.use_rew else 0) if m.use_cost else None
        self.S = R * T + P
        self.M = B * self.S
        self.BT = B * T
        M, BT = self.M, self.BT
        self.feat_ops = bool(m.add_cost_feat or m.mul_cost_feat or m.cat_cost_feat)
        Eh = self.Eh = m.head_in_dim  # 2E with cat_cost_feat (cdt.py:125)
        f = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f)  # noqa: E731
        self.st = StepState(dev, STAT_KEYS, betas=tuple(trainer_cfg["betas"]), warmup=trainer_cfg["lr_warmup_steps"])
        g: FlatGroup = m.groups["cdt"]
        self.g = g
        # static inputs
        self.states, self.actions = z(B, T, od), z(B, T, ad)
        self.returns, self.ctg, self.mask, self.costs = z(B, T), z(B, T), z(B, T), z(B, T)
        self.time_steps = torch.zeros(B, T, dtype=torch.int64, device=dev)
        # forward activations
        self.seq, self.x0, self.st_emb, self.ctg_t = z(M, E), z(M, E), z(M, 2), z(BT)
        self.xin: List[torch.Tensor] = [self.x0] + [z(M, E) for _ in range(NL)]  # input of block l / final x
        self.n1 = [z(M, E) for _ in range(NL)]
        self.st1 = [z(M, 2) for _ in range(NL)]
        self.qkv = [z(M, 3 * E) for _ in range(NL)]
        # round 6: the attention-probability dropout's keep decisions, handed from the forward to the backward launch of a
        # layer (one nibble per four keys; head widths 16 / 32) instead of 15 Philox calls per lane in the backward
        self.attn_keep = None
        kb = int(L.load().osrl_attention_keep_bytes(B, self.S, E, self.H)) if (ATTN_KEEP and not inference) else 0
        if kb > 0 and m.attention_dropout > 0:
            self.attn_keep = [torch.zeros(kb, dtype=torch.uint8, device=dev) for _ in range(NL)]
        self.o = [z(M, E) for _ in range(NL)]
        self.att = z(M, E)
        self.xmid = [z(M, E) for _ in range(NL)]
        self.n2 = [z(M, E) for _ in range(NL)]
        self.st2 = [z(M, 2) for _ in range(NL)]
        self.hpre = [z(M, 4 * E) for _ in range(NL)]
        self.h = [z(M, 4 * E) for _ in range(NL)]
        self.mo = z(M, E)
        self.out, self.st_out = z(M, E), z(M, 2)
        nh = 2 * ad if m.stochastic else ad
        self.head, self.logits, self.sp = z(BT, nh), z(BT, 2), z(BT, od)
        # backward buffers
        self.dhead, self.dlogits, self.dsp, self.ent = z(BT, nh), z(BT, 2), z(BT, od), z(4)
        self.dout = z(M, E)
        self.dxo = [z(M, E) for _ in range(NL + 1)]   # grad wrt xin[l]
        self.dxm = [z(M, E) for _ in range(NL)]       # grad wrt xmid[l]
        self.dh = z(M, 4 * E)
        self.dhpre = [z(M, 4 * E) for _ in range(NL)]
        self.dn = z(M, E)
        self.do = z(M, E)
        self.dqkv = [z(M, 3 * E) for _ in range(NL)]
        self.dseq = z(M, E)
        # cost prefix: the heads and the embedding dW read the sequence WITHOUT the leading token as [B*T, R*E] rows;
        # with the prefix the per-sample stride is not T*R*E any more, so compact copies carry that view
        self.outc = z(B * T * R, E) if P else self.out
        self.doutc = z(B * T * R, E) if P else self.dout
        self.dseqc = z(B * T * R, E) if P else self.dseq
        self.sf3 = torch.as_strided(self.outc, (B, T, E), (T * R * E, R * E, 1), (R - 2) * E)    # state-token features
        self.dsf3 = torch.as_strided(self.doutc, (B, T, E), (T * R * E, R * E, 1), (R - 2) * E)
        if self.feat_ops:  # cdt.py:243-250: the detached cost embedding (pre-LayerNorm, with its time embedding)
            self.ce3 = torch.as_strided(self.seq, (B, T, E), (self.S * E, R * E, 1), (P + self.slot_cost) * E)
            self.feat, self.dfeat = z(BT, Eh), z(BT, Eh)
        # action head (cdt.py:127-137): hidden Linear + GELU layers in front of the output layer
        self.head_hidden: List[str] = list(m.head_hidden_keys)
        self.head_out: str = m.head_out_key
        nhid = len(self.head_hidden)
        self.hh_pre = [z(BT, Eh) for _ in range(nhid)]
        self.hh = [z(BT, Eh) for _ in range(nhid)]
        self.dhh = [z(BT, Eh) for _ in range(nhid)]
        self.dhh_pre = [z(BT, Eh) for _ in range(nhid)]
        # dropout: probabilities, generator seed; gradients of the dropped branches need their own buffers
        # (the undropped gradient keeps flowing along the residual path)
        self.p_emb, self.p_attn, self.p_res = m.embedding_dropout, m.attention_dropout, m.residual_dropout
        for name, pv in (("embedding_dropout", self.p_emb), ("attention_dropout", self.p_attn),
                         ("residual_dropout", self.p_res)):
            if not (0.0 <= float(pv) < 1.0):  # p = 1 drops everything: the kernels' keep scale 1 / (1 - p) has no value
                raise ValueError(f"CDT {name} = {pv}: the dropout kernels take 0 <= p < 1")
        self.seed = int(trainer_cfg.get("seed", 0))
        if dist is not None:  # independent dropout masks per rank
            self.seed = dist.rank_seed(self.seed)
        self.datt = [z(M, E) if self.p_res > 0 else self.dxm[l] for l in range(NL)]      # grad wrt out_proj output
        self.dmo = [z(M, E) if self.p_res > 0 else self.dxo[l + 1] for l in range(NL)]   # grad wrt mlp.2 output
        self.n_parts = max(1, min(1024, (M + 31) // 32))  # ~8 rows per wave per LayerNorm-backward workgroup
        # (dgamma | dbeta) partials of every LayerNorm of the step (out_norm, norm2 / norm1 of each block, emb_norm): one
        # workspace per site, summed by ONE launch behind the last backward (osrl_layernorm_param_reduce) -- eight 5 us
        # launches inside the backward chain at C5 before
        self.ln_sites: List[str] = []
        self.ln_batch = LN_BATCH and 2 * NL + 2 <= 16
        self.ln_ws = z(2 * NL + 2 if (self.ln_batch and not inference) else 1, self.n_parts, 2 * E)
        self.clip_ws, self.clip_out = z(1024), z(4)
        self.temp_mv = z(2)
        self.counts = z(4)
        self.loss_ws = z(8 * ((BT + 1023) // 1024) + 8)
        self._graph_failed = False
        self._ln_pending: List[str] = []

        # dW plans
        tok, bt = [], []
        for l in range(NL):
            p = f"cdt.blocks.{l}."
            tok += [(self.dqkv[l], self.n1[l], p + "attention.in_proj_weight", p + "attention.in_proj_bias"),
                    (self.datt[l], self.o[l], p + "attention.out_proj.weight", p + "attention.out_proj.bias"),
                    (self.dhpre[l], self.n2[l], p + "mlp.0.weight", p + "mlp.0.bias"),
                    (self.dmo[l], self.h[l], p + "mlp.2.weight", p + "mlp.2.bias")]
        sf_ptr, af_ptr = self.outc.data_ptr() + 4 * (R - 2) * E, self.outc.data_ptr() + 4 * (R - 1) * E
        ds = self.dseqc.data_ptr()
        wb = lambda wkey: (wkey, wkey[:-len("weight")] + "bias")  # noqa: E731
        chain = self.head_hidden + [self.head_out]
        for li, key in enumerate(chain):  # (dz, a, weight, bias, ldz, lda): 0 = dense
            dz = self.dhead if li == len(chain) - 1 else self.dhh_pre[li]
            if li > 0:
                a_ptr, lda = self.hh[li - 1].data_ptr(), 0
            elif self.feat_ops:
                a_ptr, lda = self.feat.data_ptr(), 0
            else:
                a_ptr, lda = sf_ptr, R * E
            bt.append((dz.data_ptr(), a_ptr) + wb(key) + (0, lda))
        bt += [(self.dlogits.data_ptr(), af_ptr, "cdt.cost_pred_head.weight", "cdt.cost_pred_head.bias", 0, R * E),
               (self.dsp.data_ptr(), af_ptr, "cdt.state_pred_head.weight", "cdt.state_pred_head.bias", 0, R * E),
               (ds + 4 * (R - 2) * E, self.states.data_ptr(), "cdt.state_emb.weight", "cdt.state_emb.bias", R * E, 0),
               (ds + 4 * (R - 1) * E, self.actions.data_ptr(), "cdt.action_emb.weight", "cdt.action_emb.bias", R * E, 0)]
        if m.use_rew:
            bt.append((ds + 4 * self.slot_rew * E, self.returns.data_ptr(), "cdt.return_emb.weight",
                       "cdt.return_emb.bias", R * E, 0))
        if m.use_cost:
            bt.append((ds + 4 * self.slot_cost * E, self.ctg_t.data_ptr(), "cdt.cost_emb.weight", "cdt.cost_emb.bias",
                       R * E, 0))
        self._keep_bt = [self.dhead, self.dhh_pre, self.hh, self.dlogits, self.dsp, self.dseqc, self.outc]
        # an inference engine (CDT.forward, CDTBatchedRollout) never launches dW: no plans, no gradient slabs
        self.inference = bool(inference)
        self.episode_cost = z(B)
        self.p_pre = None
        if not self.inference:
            self.p_tok = DwPlan(g, tok, M, dev)
            self.p_bt = DwPlan(g, bt, BT, dev)
            self.n_splits = max(self.p_tok.n_splits, self.p_bt.n_splits)
            if P:  # the prefix token's Linear(1, E): one row per sample (row 0 of each sequence in dseq)
                self.p_pre = DwPlan(g, [(self.dseq.data_ptr(), self.episode_cost.data_ptr(), "cdt.prefix_emb.weight",
                                         "cdt.prefix_emb.bias", self.S * E, 0)], B, dev)
                self.n_splits = max(self.n_splits, self.p_pre.n_splits)
        # row splits per 1024-float chunk of the flat gradient (what osrl_reduce_slabs_counts sums): the plans' own counts
        # for the ranges they write, 1 for everything written straight into slab 0 (LayerNorm parameters, timestep rows)
        self.slab_counts = None
        if not self.inference and SLAB_COUNTS:
            cnt = [1] * ((g.n + 1023) // 1024)
            for pl in (self.p_tok, self.p_bt, self.p_pre):
                for off, ln, ns in (pl.split_ranges() if pl is not None else ()):
                    for ch in range(off // 1024, (off + max(ln, 1) - 1) // 1024 + 1):
                        cnt[ch] = max(cnt[ch], min(ns, self.n_splits))
            self.slab_counts = torch.tensor(cnt, dtype=torch.uint8, device=dev)
        # every dW plan of this engine is built: the slab epochs they were built against are recorded NOW (not at the
        # first step), so an engine that is constructed directly, never stepped and then superseded is flagged stale
        from .core import slab_epochs
        self._slab_epochs = slab_epochs(self.model)
        self._slab_probe = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.store = None
        m.repack()

    # ---- thin launch helpers -------------------------------------------------------------------
    def _P(self, key: str, bwd: bool) -> int:
        g = self.g
        return (g.pb if bwd else g.pf).data_ptr() + 4 * (g.b_off if bwd else g.f_off)[key]

    def _v(self, key: str) -> int:
        return self.g.view(key).data_ptr()

    def _lin(self, A, lda, Mrows, key, Y, ldy, bias=True, resid=None, ldr=0):
        """Y = A W^T (+ b) (+ resid) with W = parameter ``key`` [N,K]."""
        N, K = self.g.layout[key][1]
        a = A if isinstance(A, int) else A.data_ptr()
        y = Y if isinstance(Y, int) else Y.data_ptr()
        r = None if resid is None else (resid if isinstance(resid, int) else resid.data_ptr())
        bkey = key[:-len("weight")] + "bias" if key.endswith("weight") else key.replace("in_proj_weight", "in_proj_bias")
        L.check(L.load().osrl_linear(a, lda, Mrows, K, self._P(key, False), _r16(N), 0, N,
                                     self._v(bkey) if bias else None, r, ldr, y, ldy, cur_stream()), "osrl_linear")

    def _lin_dx(self, dY, ldd, Mrows, key, Y, ldy, resid=None, ldr=0):
        """Y = dY W (+ resid): input gradient of the layer with parameter ``key`` [N,K]."""
        N, K = self.g.layout[key][1]
        a = dY if isinstance(dY, int) else dY.data_ptr()
        y = Y if isinstance(Y, int) else Y.data_ptr()
        r = None if resid is None else (resid if isinstance(resid, int) else resid.data_ptr())
        L.check(L.load().osrl_linear(a, ldd, Mrows, N, self._P(key, True), _r16(K) + 16, 0, K, None, r, ldr, y, ldy,
                                     cur_stream()), "osrl_linear(dx)")

    def _ln_fwd(self, x, delta, key, xout, y, stats, drop=None):
        """``drop`` = (site, p): ``delta`` is a residual branch BEFORE its nn.Dropout -- the keep-multiplier is applied
        by this launch (osrl_layernorm_fwd_drop: same bits as osrl_dropout + the plain call, two HBM passes fewer)."""
        if drop is not None and drop[1] > 0 and FUSE_DROP:
            d = self._site(*drop)
            L.check(L.load().osrl_layernorm_fwd_drop(x.data_ptr(), delta.data_ptr(), ctypes.byref(d),
                                                     self._v(key + ".weight"), self._v(key + ".bias"),
                                                     None if xout is None else xout.data_ptr(), y.data_ptr(),
                                                     stats.data_ptr(), self.M, self.E, cur_stream()),
                    "osrl_layernorm_fwd_drop")
            return
        if drop is not None and drop[1] > 0:
            self._drop(delta, delta, drop[0], drop[1])
        L.check(L.load().osrl_layernorm_fwd(x.data_ptr(), None if delta is None else delta.data_ptr(),
                                            self._v(key + ".weight"), self._v(key + ".bias"),
                                            None if xout is None else xout.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                            self.M, self.E, cur_stream()), "osrl_layernorm_fwd")

    def _ln_bwd(self, dy, x, stats, key, dres, dx, drop=None, dx_dropped=None):
        """``drop`` = (site, p), ``dx_dropped``: also write dx * keep-multiplier of that site (the gradient entering the
        residual branch whose dropout output fed this LayerNorm's input)."""
        g = self.g
        ws, slab = self.ln_ws.data_ptr(), g.slabs.data_ptr()
        if self.ln_batch:  # this site's own workspace; the reduction waits for _ln_param_reduce()
            ws, slab = self.ln_ws[len(self._ln_pending)].data_ptr(), None
            self._ln_pending.append(key)
        if drop is not None and drop[1] > 0 and FUSE_DROP:
            d = self._site(*drop)
            L.check(L.load().osrl_layernorm_bwd_drop(
                dy.data_ptr(), x.data_ptr(), stats.data_ptr(), self._v(key + ".weight"),
                None if dres is None else dres.data_ptr(), dx.data_ptr(), dx_dropped.data_ptr(), ctypes.byref(d),
                ws, self.n_parts, self.M, self.E, slab, g.offset(key + ".weight"),
                g.offset(key + ".bias"), cur_stream()), "osrl_layernorm_bwd_drop")
            return
        L.check(L.load().osrl_layernorm_bwd(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), self._v(key + ".weight"),
                                            None if dres is None else dres.data_ptr(), dx.data_ptr(),
                                            ws, self.n_parts, self.M, self.E, slab,
                                            g.offset(key + ".weight"), g.offset(key + ".bias"), cur_stream()),
                "osrl_layernorm_bwd")
        if drop is not None and drop[1] > 0:
            self._drop(dx, dx_dropped, drop[0], drop[1])

    def _ln_param_reduce(self) -> None:
        """dgamma / dbeta of every LayerNorm whose backward ran since the last call, in one launch."""
        keys, self._ln_pending = self._ln_pending, []
        if not keys:
            return
        g = self.g
        n = len(keys)
        g_offs = (ctypes.c_int64 * n)(*[g.offset(k + ".weight") for k in keys])
        b_offs = (ctypes.c_int64 * n)(*[g.offset(k + ".bias") for k in keys])
        L.check(L.load().osrl_layernorm_param_reduce(self.ln_ws.data_ptr(), self.n_parts * 2 * self.E, n, self.n_parts,
                                                     self.E, g.slabs.data_ptr(), g_offs, b_offs, cur_stream()),
                "osrl_layernorm_param_reduce")

    # ---- dropout sites: 0 = embedding; layer l: 1+3l attention probabilities, 2+3l / 3+3l the two residual branches
    def _site(self, site: int, p: float):
        return L.DropoutT(float(p), site, self.seed, self.st.ptr)

    def _drop(self, x, y, site: int, p: float) -> None:
        d = self._site(site, p)
        L.check(L.load().osrl_dropout(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(d), cur_stream()),
                "osrl_dropout")

    def dropout_masks(self) -> Dict[str, torch.Tensor]:
        """Keep-multipliers (0 or 1/(1-p)) of the most recent train step, in tensor layout: 'emb' [B,S,E],
        'attn{l}' [B,H,S,S], 'res1_{l}' / 'res2_{l}' [B,S,E].  Regenerated from the counters (diagnostics, parity tests)."""
        B, S, E, H = self.B, self.S, self.E, self.H
        out: Dict[str, torch.Tensor] = {}
        ones = torch.ones(self.M, E, device=self.dev)
        if self.p_emb > 0:
            out["emb"] = torch.empty_like(ones)
            self._drop(ones, out["emb"], 0, self.p_emb)
            out["emb"] = out["emb"].view(B, S, E)
        for l in range(self.NL):
            if self.p_attn > 0:
                Sp = _r16(S)  # the kernels' mask layout: [B*H, S, Sp], one Philox call per 4 consecutive keys
                o1 = torch.ones(B * H, S, Sp, device=self.dev)
                raw = torch.empty_like(o1)
                self._drop(o1, raw, 1 + 3 * l, self.p_attn)
                out[f"attn{l}"] = raw[:, :, :S].reshape(B, H, S, S)
            if self.p_res > 0:
                for k, site in ((f"res1_{l}", 2 + 3 * l), (f"res2_{l}", 3 + 3 * l)):
                    t = torch.empty_like(ones)
                    self._drop(ones, t, site, self.p_res)
                    out[k] = t.view(B, S, E)
        return out

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, train: bool = False) -> None:
        """``train`` enables dropout (nn.Module.training of the reference model)."""
        m, lib, E, M, BT = self.model, L.load(), self.E, self.M, self.BT
        v = self._v
        p_emb, p_attn, p_res = (self.p_emb, self.p_attn, self.p_res) if train else (0.0, 0.0, 0.0)
        opt = lambda key, on: v(key) if on else None  # noqa: E731
        L.check(lib.osrl_cdt_embed_ln(
            self.states.data_ptr(), self.actions.data_ptr(), self.returns.data_ptr(), self.ctg.data_ptr(),
            self.episode_cost.data_ptr(), self.time_steps.data_ptr(), v("cdt.state_emb.weight"),
            v("cdt.state_emb.bias"), v("cdt.action_emb.weight"), v("cdt.action_emb.bias"),
            opt("cdt.cost_emb.weight", m.use_cost), opt("cdt.cost_emb.bias", m.use_cost),
            opt("cdt.return_emb.weight", m.use_rew), opt("cdt.return_emb.bias", m.use_rew),
            opt("cdt.prefix_emb.weight", self.P), opt("cdt.prefix_emb.bias", self.P),
            opt("cdt.timestep_emb.weight", m.time_emb), v("cdt.emb_norm.weight"), v("cdt.emb_norm.bias"), self.B, self.T,
            m.state_dim, m.action_dim, E, 1 if m.cost_transform_on else 0, 1 if m.use_rew else 0,
            1 if m.use_cost else 0, self.P, self.seq.data_ptr(), self.x0.data_ptr(), self.st_emb.data_ptr(),
            self.ctg_t.data_ptr(), cur_stream()), "osrl_cdt_embed_ln")
        if p_emb > 0:
            self._drop(self.x0, self.x0, 0, p_emb)
        for l in range(self.NL):
            p = f"cdt.blocks.{l}."
            if l == 0:
                self._ln_fwd(self.xin[0], None, p + "norm1", None, self.n1[0], self.st1[0])
            self._lin(self.n1[l], E, M, p + "attention.in_proj_weight", self.qkv[l], 3 * E)
            da = self._site(1 + 3 * l, p_attn)
            keep = self.attn_keep[l].data_ptr() if (train and self.attn_keep is not None and p_attn > 0) else None
            L.check(lib.osrl_attention_fwd_keep(self.qkv[l].data_ptr(), self.mask.data_ptr(), self.B, self.S, E, self.H,
                                                self.R, self.P, ctypes.byref(da) if p_attn > 0 else None,
                                                self.o[l].data_ptr(), keep, cur_stream()), "osrl_attention_fwd")
            self._lin(self.o[l], E, M, p + "attention.out_proj.weight", self.att, E)
            self._ln_fwd(self.xin[l], self.att, p + "norm2", self.xmid[l], self.n2[l], self.st2[l],
                         drop=(2 + 3 * l, p_res))
            self._lin(self.n2[l], E, M, p + "mlp.0.weight", self.hpre[l], 4 * E)
            L.check(lib.osrl_gelu_fwd(self.hpre[l].data_ptr(), self.h[l].data_ptr(), M * 4 * E, cur_stream()), "gelu")
            self._lin(self.h[l], 4 * E, M, p + "mlp.2.weight", self.mo, E)
            if l + 1 < self.NL:
                self._ln_fwd(self.xmid[l], self.mo, f"cdt.blocks.{l + 1}.norm1", self.xin[l + 1], self.n1[l + 1],
                             self.st1[l + 1], drop=(3 + 3 * l, p_res))
            else:
                self._ln_fwd(self.xmid[l], self.mo, "cdt.out_norm", self.xin[l + 1], self.out, self.st_out,
                             drop=(3 + 3 * l, p_res))
        R, B, T, Eh = self.R, self.B, self.T, self.Eh
        if self.P:  # out[:, 1:] (cdt.py:229-231) as a dense [B*T*R, E] matrix
            self.outc.view(B, R * T, E).copy_(self.out.view(B, self.S, E)[:, 1:])
        sf, af = self.outc.data_ptr() + 4 * (R - 2) * E, self.outc.data_ptr() + 4 * (R - 1) * E
        hin, ldh = sf, R * E
        if self.feat_ops:  # state feature (+ / * / cat) detached cost embedding, cdt.py:243-250
            f3 = self.feat.view(B, T, Eh)[..., :E]
            if m.add_cost_feat:
                torch.add(self.sf3, self.ce3, out=f3)
                if m.mul_cost_feat:
                    f3.mul_(self.ce3)
            elif m.mul_cost_feat:
                torch.mul(self.sf3, self.ce3, out=f3)
            else:
                f3.copy_(self.sf3)
            if m.cat_cost_feat:
                self.feat.view(B, T, Eh)[..., E:].copy_(self.ce3)
            hin, ldh = self.feat, Eh
        for i, key in enumerate(self.head_hidden):
            self._lin(hin, ldh, BT, key, self.hh_pre[i], Eh)
            L.check(lib.osrl_gelu_fwd(self.hh_pre[i].data_ptr(), self.hh[i].data_ptr(), BT * Eh, cur_stream()), "gelu")
            hin, ldh = self.hh[i], Eh
        self._lin(hin, ldh, BT, self.head_out, self.head, self.head.shape[1])
        self._lin(af, R * E, BT, "cdt.cost_pred_head.weight", self.logits, 2)
        self._lin(af, R * E, BT, "cdt.state_pred_head.weight", self.sp, m.state_dim)

    # ---- one full train step -------------------------------------------------------------------
    def body(self) -> None:
        m, lib, cfg, g = self.model, L.load(), self.cfg, self.g
        E, M, BT, NL = self.E, self.M, self.BT, self.NL
        st = self.st
        if self.inference:
            raise RuntimeError("this CDTEngine was built for inference (no dW plans): it cannot run a train step")
        self._ln_pending = []  # (a step that raised half way must not leave sites behind)
        st.tick()
        if self.store is not None:  # draw the minibatch of windows on device (SequenceDataset, dataset.py:749-787)
            self.store.gather(self.states, self.actions, self.returns, self.ctg, self.time_steps, self.mask,
                              self.episode_cost, self.costs, st.ptr)
        self.forward(train=True)
        counts, world = None, 1
        big = BT > 1024  # multi-workgroup loss: the normalisers are needed before the per-token gradients
        if self.dist is not None or big:  # count-normalisers (over the GLOBAL batch)
            L.check(lib.osrl_cdt_mask_counts(self.mask.data_ptr(), BT, self.counts.data_ptr(), cur_stream()), "counts")
            counts = self.counts.data_ptr()
        if self.dist is not None:
            self.dist.all_reduce_(self.counts)
            world = self.dist.world
        L.check(lib.osrl_cdt_loss(self.head.data_ptr(), self.logits.data_ptr(), self.sp.data_ptr(),
                                  self.actions.data_ptr(), self.states.data_ptr(), self.mask.data_ptr(),
                                  self.costs.data_ptr(), self.B, self.T, m.state_dim, m.action_dim,
                                  1 if m.stochastic else 0, 1 if cfg["no_entropy"] else 0,
                                  m.log_temperature.data_ptr() if m.stochastic else None, cfg["loss_cost_weight"],
                                  cfg["loss_state_weight"], cfg["learning_rate"], cfg["lr_warmup_steps"], st.ptr,
                                  counts, world, self.dhead.data_ptr(), self.dlogits.data_ptr(), self.dsp.data_ptr(),
                                  st.stats.data_ptr(), self.ent.data_ptr(), self.loss_ws.data_ptr() if big else None,
                                  cur_stream()), "osrl_cdt_loss")
        # ---- backward: heads -> dout (only the state / action token rows are non-zero)
        R, B, T, Eh = self.R, self.B, self.T, self.Eh
        # dout / doutc: the rows of the return / cost (/ prefix) tokens carry no gradient from the heads and are never
        # written by anything -- they keep the zeros they were allocated with; the state / action rows are fully
        # rewritten below every step.  (Round 2 re-zeroed the 84 MB buffer per step: an aten fill launch, 0.3 ms at C5.)
        d_sf, d_af = self.doutc.data_ptr() + 4 * (R - 2) * E, self.doutc.data_ptr() + 4 * (R - 1) * E
        chain = self.head_hidden + [self.head_out]
        for li in range(len(chain) - 1, -1, -1):  # output layer first, then the hidden Linear + GELU layers
            dz, n = (self.dhead, self.dhead.shape[1]) if li == len(chain) - 1 else (self.dhh_pre[li], Eh)
            if li > 0:
                self._lin_dx(dz, n, BT, chain[li], self.dhh[li - 1], Eh)
                L.check(lib.osrl_gelu_bwd(self.dhh[li - 1].data_ptr(), self.hh_pre[li - 1].data_ptr(),
                                          self.dhh_pre[li - 1].data_ptr(), BT * Eh, cur_stream()), "gelu_bwd")
            elif self.feat_ops:
                self._lin_dx(dz, n, BT, chain[li], self.dfeat, Eh)
            else:
                self._lin_dx(dz, n, BT, chain[li], d_sf, R * E)
        if self.feat_ops:  # the cost embedding is detached: only the state-token path carries gradient
            g3 = self.dfeat.view(B, T, Eh)[..., :E]
            if m.mul_cost_feat:
                torch.mul(g3, self.ce3, out=self.dsf3)
            else:
                self.dsf3.copy_(g3)
        self._lin_dx(self.dlogits, 2, BT, "cdt.cost_pred_head.weight", d_af, R * E)
        self._lin_dx(self.dsp, m.state_dim, BT, "cdt.state_pred_head.weight", d_af, R * E, resid=d_af, ldr=R * E)
        if self.P:
            self.dout.view(B, self.S, E)[:, 1:].copy_(self.doutc.view(B, R * T, E))
        # (each LayerNorm backward also leaves dx * keep-mask of the residual branch that fed its input: the gradient the
        # branch's last Linear needs -- osrl_layernorm_bwd_drop)
        self._ln_bwd(self.dout, self.xin[NL], self.st_out, "cdt.out_norm", None, self.dxo[NL],
                     drop=(3 + 3 * (NL - 1), self.p_res), dx_dropped=self.dmo[NL - 1])
        for l in range(NL - 1, -1, -1):
            p = f"cdt.blocks.{l}."
            self._lin_dx(self.dmo[l], E, M, p + "mlp.2.weight", self.dh, 4 * E)
            L.check(lib.osrl_gelu_bwd(self.dh.data_ptr(), self.hpre[l].data_ptr(), self.dhpre[l].data_ptr(),
                                      M * 4 * E, cur_stream()), "gelu_bwd")
            self._lin_dx(self.dhpre[l], 4 * E, M, p + "mlp.0.weight", self.dn, E)
            self._ln_bwd(self.dn, self.xmid[l], self.st2[l], p + "norm2", self.dxo[l + 1], self.dxm[l],
                         drop=(2 + 3 * l, self.p_res), dx_dropped=self.datt[l])
            self._lin_dx(self.datt[l], E, M, p + "attention.out_proj.weight", self.do, E)
            da = self._site(1 + 3 * l, self.p_attn)
            keep = self.attn_keep[l].data_ptr() if (self.attn_keep is not None and self.p_attn > 0) else None
            L.check(lib.osrl_attention_bwd_keep(self.qkv[l].data_ptr(), self.mask.data_ptr(), self.do.data_ptr(), self.B,
                                                self.S, E, self.H, self.R, self.P,
                                                ctypes.byref(da) if self.p_attn > 0 else None,
                                                self.dqkv[l].data_ptr(), keep, cur_stream()), "attn_bwd")
            self._lin_dx(self.dqkv[l], 3 * E, M, p + "attention.in_proj_weight", self.dn, E)
            if l > 0:
                self._ln_bwd(self.dn, self.xin[l], self.st1[l], p + "norm1", self.dxm[l], self.dxo[l],
                             drop=(3 + 3 * (l - 1), self.p_res), dx_dropped=self.dmo[l - 1])
            else:
                self._ln_bwd(self.dn, self.xin[l], self.st1[l], p + "norm1", self.dxm[l], self.dxo[l])
        if self.p_emb > 0:
            self._drop(self.dxo[0], self.dxo[0], 0, self.p_emb)
        self._ln_bwd(self.dxo[0], self.seq, self.st_emb, "cdt.emb_norm", None, self.dseq)
        self._ln_param_reduce()
        # ---- parameter gradients
        if m.time_emb:
            te_off, (te_rows, _) = g.layout["cdt.timestep_emb.weight"]
            g.slabs[0, te_off:te_off + te_rows * E].zero_()
            L.check(lib.osrl_cdt_timestep_scatter(self.dseq.data_ptr(), self.time_steps.data_ptr(), self.B, self.T,
                                                  self.R, self.P, E, g.slabs.data_ptr() + 4 * te_off, cur_stream()),
                    "te_scatter")
        if self.P:
            self.dseqc.view(self.B, self.R * self.T, E).copy_(self.dseq.view(self.B, self.S, E)[:, 1:])
            self.p_pre.launch()
        self.p_tok.launch()
        self.p_bt.launch()
        g.cur_splits = self.n_splits
        # ---- clip_grad_norm_ + AdamW (cdt.py:396-400)
        if self._slab_probe is not None:  # tests: the complete slabs of a real step, before they are summed in place
            self._slab_probe(g.slabs, self.n_splits, self.slab_counts)
        if self.slab_counts is not None:
            L.check(lib.osrl_reduce_slabs_counts(g.slabs.data_ptr(), g.slabs.data_ptr(), self.slab_counts.data_ptr(), g.n,
                                                 g.n, cur_stream()), "osrl_reduce_slabs_counts")
        else:
            L.check(lib.osrl_reduce_slabs(g.slabs.data_ptr(), g.slabs.data_ptr(), g.cur_splits, g.n, g.n, cur_stream()),
                    "osrl_reduce_slabs")
        g.cur_splits = 1
        if self.dist is not None:  # ONE all-reduce of the flat gradient; the clip norm is of the reduced gradient
            self.dist.all_reduce_(g.slabs[0])
            self.dist.all_reduce_(self.ent)
        clip = cfg["clip_grad"]
        gscale = None
        if clip is not None:
            L.check(lib.osrl_clip_grad_scale(g.slabs.data_ptr(), g.n, float(clip), self.clip_ws.data_ptr(), 512,
                                             self.clip_out.data_ptr(), cur_stream()), "osrl_clip_grad_scale")
            gscale = self.clip_out
        g.adam_step(cfg["learning_rate"], st.ptr, betas=tuple(cfg["betas"]), weight_decay=cfg["weight_decay"],
                    gscale=gscale, polyak=False)
        if m.stochastic:  # cdt.py:402-407
            L.check(lib.osrl_cdt_temperature_step(m.log_temperature.data_ptr(), self.temp_mv.data_ptr(),
                                                  self.ent.data_ptr(), float(m.target_entropy), 1e-4, 0.9, 0.999, 1e-8,
                                                  st.ptr, cur_stream()), "osrl_cdt_temperature_step")
        if self.dist is not None:
            self.dist.all_reduce_(st.stats)

    def load_batch(self, states, actions, returns, costs_return, time_steps, mask, costs, episode_cost=None) -> None:
        pairs = [(self.states, states), (self.actions, actions), (self.returns, returns), (self.ctg, costs_return),
                 (self.time_steps, time_steps), (self.mask, mask), (self.costs, costs)]
        if episode_cost is not None:  # only the cost-prefix variant reads it (cdt.py:207-213)
            pairs.append((self.episode_cost, episode_cost))
        elif self.P:
            raise ValueError("cost_prefix=True: episode_cost is an input of the model")
        load_into(pairs)

    def attach_store(self, store) -> None:
        if store is not None and self.dist is not None:
            store.set_rank(self.dist.rank)  # each rank samples its own windows
        self.store = store
        self.graph = None

    def step_store(self, use_graph: bool = True) -> None:
        """One train step on windows sampled on device from the attached SequenceStore."""
        check_plans_current(self)
        assert self.store is not None
        self._go(use_graph)

    def step(self, states, actions, returns, costs_return, time_steps, mask, costs, use_graph: bool = True,
             episode_cost=None) -> None:
        check_plans_current(self)
        if self.store is not None:
            raise RuntimeError("a SequenceStore is attached: call step_store()")
        self.load_batch(states, actions, returns, costs_return, time_steps, mask, costs, episode_cost)
        self._go(use_graph)

    def _go(self, use_graph: bool) -> None:
        if use_graph and not self._graph_failed:
            if self.graph is None:
                ok = True
                try:
                    self._capture()
                except Exception as e:  # pragma: no cover - depends on the RCCL build
                    if self.dist is None:
                        raise
                    import warnings
                    warnings.warn(f"hipGraph capture of the data-parallel CDT step failed ({e!r}); running eagerly")
                    ok = False
                if self.dist is not None and not self.dist.all_agree(ok, self.dev):
                    self._graph_failed, self.graph = True, None  # every rank runs eagerly, or none does
            if self.graph is not None:
                self.graph.replay()
                self.st.host_step += 1
                return
        self.body()

    def _capture(self) -> None:
        m, g = self.model, self.g
        snap = (g.p.clone(), g.m.clone(), g.v.clone(), self.st.state.clone(), self.st.stats.clone(),
                self.st.ring.clone(), self.st.host_step, m.log_temperature.clone() if m.stochastic else None,
                self.temp_mv.clone())
        try:  # warm-up + capture both run a real step: the snapshot goes back even when the capture is refused
            gr, self._arena = capture_step(self.st.state.device, self.body, self.body)
        finally:
            torch.cuda.synchronize()
            g.p.copy_(snap[0]); g.m.copy_(snap[1]); g.v.copy_(snap[2])
            self.st.state.copy_(snap[3]); self.st.stats.copy_(snap[4]); self.st.ring.copy_(snap[5])
            self.st.host_step = snap[6]
            if m.stochastic:
                m.log_temperature.copy_(snap[7])
            self.temp_mv.copy_(snap[8])
            m.repack()
        self.graph = gr
