#!/usr/bin/env python3
"""The two N*B-row forward launches of the CPQ step, exactly as the step (and bench.py's roofline()) issues them -- the
engine's own descriptors, row maps (observation r % B, sampled action r) and buffers -- repeated in isolation, for the
`rocprofv3 --pmc` passes of tools/gpu_r6c_pmc.sh.  The SHIPPED forms: the encoder launch on its shared-observation tiles
(plan.ood_share), the target cost critics on the row list the step's select launch left (plan.ood_rows) or, without that plan,
on all N*B rows of their shared-observation tiles.    usage: pmc_nb.py c2|c4 [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from osrl_amd import _lib as L  # noqa: E402
from osrl_amd.engine import glue as G  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
wl = bench.Workload(name, dev, 0, 1, None, n_store=1 << 16, use_graph=False)
eng = wl.eng
wl.step()  # one eager step: a sampled minibatch and the N*B sampled actions are in the buffers the launches read
torch.cuda.synchronize()
print(f"{name}: ood_rows={bool(eng.ood_rows)} selected rows {int(eng.ood_count[0].item())} share_k16 enc {eng.pre_enc} cost {eng.pre_cost}", file=sys.stderr)
Lz = eng.model.latent_dim
for _ in range(iters):
    eng.r_enc_ood.forward(eng.obs, eng.sampled, map0=L.MAP_MOD, div0=eng.B, tail=G.vae_kl_tail(Lz, eng.kl), share_k16=eng.pre_enc)
torch.cuda.synchronize()
for _ in range(iters):
    if eng.ood_rows:
        eng.r_costold_ood.forward(eng.obs, eng.sampled, map0=L.MAP_MOD, div0=eng.B, row_list=eng.ood_list, n_rows_dev=eng.ood_count)
    else:
        eng.r_costold_ood.forward(eng.obs, eng.sampled, map0=L.MAP_MOD, div0=eng.B, share_k16=eng.pre_cost)
torch.cuda.synchronize()
