#!/usr/bin/env python3
"""What does a 2048-row chain launch lose beside an N*B-row launch?  Inside a hipGraph (two branches, as the train step
runs; eager two-stream launches are dominated by the runtime's cross-queue dependency handling): branch A = NA
chain launches (4 critics on (obs, a): the actor-phase forward of CPQ), branch B = NB N*B-row encoder launches of
``rows`` rows.  Prints the replay time of A alone, B alone and both.  The N*B-row launch comes from the library given
as argv[1] (a probe build of csrc/mlp.hip: OSRL_EXP_NO_BLOAD / OSRL_EXP_NO_MFMA), the chain launch from the product build.
Usage: [PROBE_NB_ROWS=20480] [PROBE_NB=3] [PROBE_NA=9] [PROBE_ORDER=side_first] [PROBE_SIDE=quantile]
       python tools/corun_probe.py [variant.so]
A variant library = csrc/mlp.hip compiled with -DOSRL_EXP_NO_BLOAD / -DOSRL_EXP_NO_MFMA, linked with the product build's
other objects (osrl_amd/lib/obj/*.o).  Results of round 2: profiles/r2_corun_probe.txt."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from osrl_amd import _lib as L  # noqa: E402
from osrl_amd.engine.core import MlpRun  # noqa: E402
from tools.kbench import mk  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    var = C.CDLL(sys.argv[1]) if len(sys.argv) > 1 else L.load()
    var.osrl_mlp_forward.argtypes = [C.POINTER(L.MlpT), C.POINTER(L.RowsT), C.POINTER(L.ActsT), C.c_void_p]
    var.osrl_mlp_forward.restype = C.c_int
    _, dq = mk(4, [78, 256, 256, 1], ["relu", "relu", "id"], dev, 16)
    _, de = mk(1, [78, 400, 400, 8], ["relu", "relu", "id"], dev, 80)
    B, NB = 2048, int(os.environ.get("PROBE_NB_ROWS", "20480"))
    NA, NBL = int(os.environ.get("PROBE_NA", "9")), int(os.environ.get("PROBE_NB", "3"))
    obs, act = torch.randn(B, 76, device=dev), torch.randn(B, 2, device=dev)
    obs2, act2 = torch.randn(NB, 76, device=dev), torch.randn(NB, 2, device=dev)
    chain = MlpRun(dq, B, True, dev)
    nb = MlpRun(de, NB, False, dev)
    r = L.RowsT()
    r.rows, r.d0, r.map0, r.div0, r.d1, r.map1, r.div1 = NB, 76, L.MAP_ID, 1, 2, L.MAP_ID, 1
    r.src0, r.src1 = obs2.data_ptr(), act2.data_ptr()
    side = torch.cuda.Stream()
    qx, qout = torch.randn(20480, device=dev).abs(), torch.zeros(4, device=dev)

    def body(a: bool, b: bool):
        cur = torch.cuda.current_stream()
        order = os.environ.get("PROBE_ORDER", "main_first")

        def side_part():
            if os.environ.get("PROBE_SIDE") == "quantile":  # a single-workgroup kernel with 136 B of LDS instead
                from osrl_amd.engine import glue as G
                with torch.cuda.stream(side):
                    for _ in range(NBL * 3):
                        G.quantile(qx, qx.numel(), 0.75, qout)
                return
            for _ in range(NBL):
                rc = var.osrl_mlp_forward(C.byref(nb.fwd_c), C.byref(r), C.byref(nb.acts_c), side.cuda_stream)
                assert rc == 0, rc

        if b:
            side.wait_stream(cur)
            if order == "side_first":
                side_part()
        if a:
            for _ in range(NA):
                chain.forward(obs, act)
        if b:
            if order != "side_first":
                side_part()
            cur.wait_stream(side)

    def timed(a, b, iters=20):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body(a, b)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                body(a, b)
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    ta, tb, tab = timed(True, False), timed(False, True), timed(True, True)
    print(f"graph replay: {NA} chain launches {ta:.0f} us | {NBL} N*B-row launches ({NB} rows) {tb:.0f} us | both "
          f"{tab:.0f} us   (sum {ta + tb:.0f}, max {max(ta, tb):.0f})")


if __name__ == "__main__":
    main()
