#!/usr/bin/env python3
"""Round 5 debug: all-CU VAE launches -- which of (eager, graph) x (arena on/off) agree bit for bit."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cases import Case
from gpu_util import build_gpu, gpu_batch, gpu_step
from test_gpu_train_step import FULL_CASES

def run(use_graph, steps=3):
    c = Case("cpq_c2_full", episode_len=1000, **FULL_CASES["cpq_c2_full"])
    m, tr, lg = build_gpu(c, stats_mode="none", use_graph=use_graph)
    b = gpu_batch(c)
    for s in range(steps):
        gpu_step(tr, c, b, s, with_noise=False)
    torch.cuda.synchronize()
    eng = m._engine
    extra = {"z": eng.z.clone(), "enc_h0": eng.r_enc.h[0][0].clone(), "enc_h1": eng.r_enc.h[0][1].clone(),
             "head": eng.r_enc.y[0].clone(), "dec_h0": eng.r_dec.h[0][0].clone(), "dec_h1": eng.r_dec.h[0][1].clone(),
             "u": eng.r_dec.y[0].clone(), "dec_dz0": eng.r_dec.dz[0][0].clone(), "dec_dz1": eng.r_dec.dz[0][1].clone(),
             "dec_dz2": eng.r_dec.dz[0][2].clone(), "enc_dz0": eng.r_enc.dz[0][0].clone(),
             "enc_dz1": eng.r_enc.dz[0][1].clone(), "enc_dz2": eng.r_enc.dz[0][2].clone(), "enc_x": eng.r_enc.x.clone(),
             "dec_x": eng.r_dec.x.clone()}
    return {k: v.clone() for k, v in m.state_dict().items()}, extra

def diff(a, b, tag):
    bad = [(k, float((a[k] - b[k]).abs().max())) for k in a if not torch.equal(a[k], b[k])]
    print(tag, "identical" if not bad else f"{len(bad)} tensors differ, e.g. {bad[:6]}")

for steps in (1, 3):
    e1, x1 = run(False, steps); e2, x2 = run(False, steps); g1, y1 = run(True, steps); g2, y2 = run(True, steps)
    print(f"--- {steps} step(s), OSRL_VAE_NS={os.environ.get('OSRL_VAE_NS')} OSRL_ARG_ARENA={os.environ.get('OSRL_ARG_ARENA')}")
    diff(e1, e2, "eager vs eager params:"); diff(g1, g2, "graph vs graph params:"); diff(e1, g1, "eager vs graph params:")
    diff(x1, x2, "eager vs eager buffers:"); diff(y1, y2, "graph vs graph buffers:"); diff(x1, y1, "eager vs graph buffers:")
