"""Debug: where does the row-list forward differ from the full forward?"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from osrl_amd import _lib as L
from osrl_amd.engine.core import FlatGroup, LayerRef, MlpRun, NetDesc
dev = torch.device("cuda:0")
E, dims, acts, rows, div0 = 2, [78, 256, 256, 1], ["relu", "relu", "id"], 20480, 2048
rs = np.random.RandomState(7)
grp = FlatGroup("t", dev)
for e in range(E):
    for l in range(len(dims) - 1):
        grp.add(f"{e}.{l}.w", (dims[l + 1], dims[l])); grp.mark_weight(f"{e}.{l}.w"); grp.add(f"{e}.{l}.b", (dims[l + 1],))
grp.finalize()
refs = []
for e in range(E):
    rr = []
    for l in range(len(dims) - 1):
        kk = 1 / math.sqrt(dims[l])
        W, b = grp.view(f"{e}.{l}.w"), grp.view(f"{e}.{l}.b")
        W.copy_(torch.tensor(rs.uniform(-kk, kk, W.shape), dtype=torch.float32)); b.copy_(torch.tensor(rs.uniform(-kk, kk, b.shape), dtype=torch.float32))
        rr.append(LayerRef(W, b, grp, f"{e}.{l}.w", f"{e}.{l}.b"))
    refs.append(rr)
grp.repack()
src0 = torch.tensor(rs.randn(div0, dims[0] - 2), dtype=torch.float32, device=dev)
src1 = torch.tensor(rs.randn(rows, 2), dtype=torch.float32, device=dev)
desc = NetDesc(refs, acts, 1.0); desc.c.tile_rows = 80
full = MlpRun(desc, rows, False, dev)
y_full = full.forward(src0, src1, map0=L.MAP_MOD, div0=div0).clone()
y_full2 = full.forward(src0, src1, map0=L.MAP_MOD, div0=div0).clone()
print("full vs full again equal:", torch.equal(y_full, y_full2))
sel = MlpRun(desc, rows, False, dev)
for name, lst in (("identity", torch.arange(rows, dtype=torch.int32, device=dev)),
                  ("shift by 1", (torch.arange(rows, dtype=torch.int32, device=dev) + 1) % rows),
                  ("shift by 16", (torch.arange(rows, dtype=torch.int32, device=dev) + 16) % rows),
                  ("shift by 80", (torch.arange(rows, dtype=torch.int32, device=dev) + 80) % rows)):
    cnt = torch.tensor([rows, 0, 0, 0], dtype=torch.int32, device=dev)
    y = sel.forward(src0, src1, map0=L.MAP_MOD, div0=div0, row_list=lst, n_rows_dev=cnt)
    torch.cuda.synchronize()
    d = (y - y_full[:, lst.long()]).abs()
    bad = (d > 0).nonzero()
    print(name, "max diff", d.max().item(), "n differing", bad.shape[0], "first", bad[:5].tolist())
os.environ["OSRL_NB_HEAD"] = "0"
y_h0 = full.forward(src0, src1, map0=L.MAP_MOD, div0=div0).clone()
lst = (torch.arange(rows, dtype=torch.int32, device=dev) + 1) % rows
cnt = torch.tensor([rows, 0, 0, 0], dtype=torch.int32, device=dev)
y = sel.forward(src0, src1, map0=L.MAP_MOD, div0=div0, row_list=lst, n_rows_dev=cnt)
torch.cuda.synchronize()
print("head as a layer, shift by 1: max diff", (y - y_h0[:, lst.long()]).abs().max().item())
