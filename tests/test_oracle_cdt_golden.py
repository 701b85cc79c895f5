"""Pin the CDT oracle (oracle/cdt_oracle.py) against vectors captured from the reference."""
import numpy as np
import pytest

from cases import CDT_CASES, make_cdt_batch, make_cdt_drop, make_cdt_params
from oracle.cdt_oracle import OracleCDT
from oracle_util import load_golden


def build_cdt_oracle(c, dtype=np.float32):
    return OracleCDT(make_cdt_params(c), seq_len=c.T, num_heads=c.heads, num_layers=c.layers,
                     cost_transform=c.cost_transform, stochastic=c.stochastic, init_temperature=0.1,
                     target_entropy=-c.ad, learning_rate=c.lr, weight_decay=c.wd, clip_grad=c.clip,
                     lr_warmup_steps=c.warmup, loss_cost_weight=c.cost_w, loss_state_weight=c.state_w,
                     time_emb=c.time_emb, use_rew=c.use_rew, use_cost=c.use_cost, add_cost_feat=c.add_cost_feat,
                     mul_cost_feat=c.mul_cost_feat, cat_cost_feat=c.cat_cost_feat, action_head_layers=c.head_layers,
                     cost_prefix=c.cost_prefix, dtype=dtype)


@pytest.mark.parametrize("name", list(CDT_CASES))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cdt_oracle_matches_reference(name, dtype):
    c = CDT_CASES[name]
    g = load_golden(name)
    keys = [str(k) for k in g["stat_keys"]]
    o = build_cdt_oracle(c, dtype)
    b = make_cdt_batch(c)
    for s in range(c.steps):
        st = o.train_one_step(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                              b["episode_cost"], b["costs"], drop=make_cdt_drop(c, s) if c.dropout > 0 else None)
        ref = dict(zip(keys, g["stats"][s]))
        tol = 1e-5 if s == 0 else 1e-4
        for k in keys:
            assert abs(st[k] - ref[k]) <= tol * max(1.0, abs(ref[k])), (name, s, k, st[k], ref[k])
        if f"s{s + 1}/log_temperature" in g:
            assert abs(o.log_temperature - float(g[f"s{s + 1}/log_temperature"])) < 1e-6
        for k, v in o.p.items():
            if f"p{s + 1}/{k}" in g:
                np.testing.assert_allclose(v, g[f"p{s + 1}/{k}"], rtol=0, atol=2e-5, err_msg=f"{name} step {s+1} {k}")
            elif f"p{s + 1}/smp/{k}" in g:
                np.testing.assert_allclose(v.reshape(-1)[::97], g[f"p{s + 1}/smp/{k}"], rtol=0, atol=2e-5)
    a = o.act_mean(b["states"], b["actions"], b["returns"], b["costs_return"], b["time_steps"], b["mask"],
                   b["episode_cost"])
    np.testing.assert_allclose(a, g["act"], rtol=0, atol=1e-4)
