"""Premise check for a CU-partitioned step: a 2048-row (latency-chain) launch beside the N*B-row launch, with the two
on streams restricted to disjoint CU sets (hipExtStreamCreateWithCUMask) vs unrestricted."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import mk
from osrl_amd.engine.core import MlpRun

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(((bits >> (32 * w + b)) & 1) << b for b in range(32)) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


_, d_chain = mk(4, [78, 256, 256, 1], ["relu", "relu", "id"], dev)
_, d_enc = mk(1, [78, 400, 400, 8], ["relu", "relu", "id"], dev, 80)
_, d_loop = mk(2, [78, 256, 256, 1], ["relu", "relu", "id"], dev)
x0 = torch.randn(2048, 76, device=dev); x1 = torch.randn(2048, 2, device=dev)
X0 = torch.randn(20480, 76, device=dev); X1 = torch.randn(20480, 2, device=dev)
chain = MlpRun(d_chain, 2048, False, dev)
enc = MlpRun(d_enc, 20480, False, dev, tile_rows=80)
loop = MlpRun(d_loop, 20480, False, dev, wg_cap=512)


def run(sa, sb, big, n_big=20, label=""):
    """big kernel n_big times on sb; chain kernel back-to-back on sa for as long as sb is busy."""
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(sb):
        e[0].record()
        for _ in range(n_big):
            big()
        e[1].record()
    n_chain = 0
    with torch.cuda.stream(sa):
        e[2].record()
        while not e[1].query() and n_chain < 2000:
            chain.forward(x0, x1)
            n_chain += 1
        e[3].record()
    torch.cuda.synchronize()
    print(f"{label:44s} big {e[0].elapsed_time(e[1]) * 1e3 / n_big:7.1f} us/launch | chain {e[2].elapsed_time(e[3]) * 1e3 / max(n_chain, 1):7.1f} us/launch ({n_chain})")


full = (1 << 256) - 1
lo, hi = (1 << 128) - 1, ((1 << 128) - 1) << 128
even = sum(1 << i for i in range(0, 256, 2)); odd = even << 1
sets = {"unmasked": (full, full), "lo128/hi128": (lo, hi), "even/odd": (even, odd),
        "chain 64 / big 192": ((1 << 64) - 1, full ^ ((1 << 64) - 1))}
for nm, (ma, mb) in sets.items():
    sa, sb = masked_stream(ma), masked_stream(mb)
    for bn, big in (("enc80", lambda: enc.forward(X0, X1)), ("loop", lambda: loop.forward(X0, X1))):
        # solo timings on the masked streams
        torch.cuda.synchronize()
        t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sb):
            big(); t[0].record()
            for _ in range(10):
                big()
            t[1].record()
        torch.cuda.synchronize()
        solo_big = t[0].elapsed_time(t[1]) * 100
        with torch.cuda.stream(sa):
            chain.forward(x0, x1); t[0].record()
            for _ in range(50):
                chain.forward(x0, x1)
            t[1].record()
        torch.cuda.synchronize()
        solo_chain = t[0].elapsed_time(t[1]) * 20
        run(sa, sb, big, label=f"{nm} {bn} (solo big {solo_big:.1f}, chain {solo_chain:.1f})")
