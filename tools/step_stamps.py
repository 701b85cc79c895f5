"""Per-phase wall-clock stamps of the one-launch BC step (csrc/mlp.hip mlp_step_kernel under -DOSRL_STEP_STAMPS):
   bash tools/build_stamps_lib.sh; OSRL_LIB=osrl_amd/lib/libosrl_stamps.so python tools/step_stamps.py [B] [hidden]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from osrl_amd import _lib as L
from osrl_amd.algorithms import BC
from osrl_amd.common.replay import ReplayStore, synthetic_transitions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
torch.manual_seed(0)
m = BC(8, 2, 1.0, [H, H], 300, device=dev)
m.setup_optimizers(1e-3)
e = m.engine(B)
e.attach_replay(ReplayStore(synthetic_transitions(100000, 8, 2, seed=1), dev, seed=3))
NAMES = ["start", "t_old", "gather", "fwd", "mse", "bwd", "arrive", "(loss)", "spin", "acquire", "dW", "adam", "done"]
for mode in ("eager", "graph"):
    for _ in range(30):
        e.step_replay(use_graph=mode == "graph")
    torch.cuda.synchronize()
    out = np.zeros((L.STEP_MAX_WG, 16), np.int64)
    fn = L.load().osrl_debug_step_stamps
    fn.argtypes = [C.c_void_p]
    assert fn(out.ctypes.data) == 0
    n_work = e._step_work.numel() // 4
    n_wg = max((B + 15) // 16, n_work)
    st = out[:n_wg, :13].astype(np.float64) * 0.01  # us
    t0 = st[:, 0].min()
    print(f"== {mode}: B={B} hidden={H} n_tiles={(B + 15) // 16} n_work={n_work} (T={e._step_T}); us since the first workgroup's start")
    print("wg   " + " ".join(f"{n:>8s}" for n in NAMES))
    for w in list(range(min(n_wg, 4))) + list(range(max(n_wg - 3, 4), n_wg)):
        print(f"{w:3d}  " + " ".join(f"{(x - t0):8.2f}" if x > 0 else "       -" for x in st[w]))
    last = st[:, 12].max() - t0
    print(f"kernel span (first start -> last done): {last:.2f} us; mean per phase over the row tiles:")
    nt = (B + 15) // 16
    d = np.diff(st[:nt], axis=1)
    print("     " + " ".join(f"{x:8.2f}" for x in [0.0] + list(d.mean(0))))
    # per-layer stamps inside the forward / backward bodies (csrc/mlp_common.h PHASE_STAMP / BWD_STAMP under OSRL_STEP_STAMPS)
    ph = np.zeros((L.STEP_MAX_WG, 2, 16), np.int64)
    fp = getattr(L.load(), "osrl_debug_step_phases", None)
    if fp is not None:
        fp.argtypes = [C.c_void_p]
        assert fp(ph.ctypes.data) == 0
        ph = ph[:nt].astype(np.float64) * 0.01
        FW = ["start", "staged"] + [f"L{l}:{n}" for l in range(3) for n in ("kloop", "barrier", "epilog", "saved")]
        BW = ["start", "dZstaged"] + [f"L{l}:{n}" for l in (2, 1) for n in ("kloop", "barrier", "stored")]
        for name, arr, names in (("forward", ph[:, 0], FW), ("backward", ph[:, 1], BW)):
            k = len(names)
            a = arr[:, :k]
            seg = np.diff(a, axis=1)
            print(f"  {name} body, us per phase (mean over the {nt} row tiles; body total {np.mean(a[:, k - 1] - a[:, 0]):.2f} us):")
            print("    " + " ".join(f"{n:>10s}" for n in names[1:]))
            print("    " + " ".join(f"{x:10.2f}" for x in seg.mean(0)))
