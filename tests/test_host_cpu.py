"""CPU-side checks of the host layer: the C-ABI library loads and exports every symbol the header
declares, ctypes struct layouts match the C structs, parameter layout / state_dict keys mirror the
reference, and the product path refuses to run without a HIP device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from osrl_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "osrl_amd.h")).read()
    names = set(re.findall(r"\b(osrl_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/osrl_amd.h but not exported"
    assert set(_lib.PROTOTYPES) | {"osrl_version"} == names, set(_lib.PROTOTYPES) ^ (names - {"osrl_version"})
    assert b"gfx950" in lib.osrl_version()


def test_ctypes_struct_sizes_match_c():
    import ctypes as C
    from osrl_amd import _lib as L
    assert C.sizeof(L.MlpT) == 56 + 3 * 8 * 4 * 8 and C.sizeof(L.PackEntryT) == 32
    assert C.sizeof(L.RowsT) == 72 and C.sizeof(L.ActsT) == 8 + 8 * 4 * 8
    assert C.sizeof(L.GradsT) == 64 + 256 + 64 + 8 and C.sizeof(L.DwEntryT) == 48 and C.sizeof(L.StepStateT) == 24


def test_no_cpu_fallback():
    from osrl_amd.algorithms import CPQ
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CPQ(5, 2, 1.0, device="cpu")
    if not torch.cuda.is_available():
        from osrl_amd.engine.core import cur_stream
        with pytest.raises(RuntimeError):
            cur_stream()


def test_state_dict_layout_matches_reference_goldens():
    import osrl_amd.engine.core as core
    from cases import CASES
    from gpu_util import build_gpu
    from oracle_util import load_golden
    core.LAYOUT_ONLY_OK = True
    try:
        for name in ("bc_small", "cpq_small", "cpq_odd", "bcql_small"):
            c = CASES[name]
            m, tr, lg = build_gpu(c, device="cpu")
            g = load_golden(name)
            want = {k[3:]: g[k].shape for k in g.files if k.startswith("p1/")}
            got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            assert got == want
            # parameters are views into the flat groups; packed heads are adjacent
            for gname, grp in m.groups.items():
                for key, (off, shape) in grp.layout.items():
                    assert off % 4 == 0 or key.endswith(("log_std_layer.weight", "log_std_layer.bias",
                                                         "log_std.weight", "log_std.bias"))
            eng = m.engine(c.B)  # builds every buffer / plan (no launches)
            assert eng.B == c.B
    finally:
        core.LAYOUT_ONLY_OK = False


def test_same_seed_same_init_as_torch_linear_order():
    """Construction order mirrors the reference, so the global torch seed reproduces nn.Linear inits."""
    import osrl_amd.engine.core as core
    from osrl_amd.algorithms import BC
    core.LAYOUT_ONLY_OK = True
    try:
        torch.manual_seed(7)
        m = BC(8, 2, 1.0, [16, 16], device="cpu")
        torch.manual_seed(7)
        ref = [torch.nn.Linear(8, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 2)]
        for i, l in enumerate(ref):
            assert torch.equal(m.state_dict()[f"actor.pi.{2 * i}.weight"], l.weight.data)
    finally:
        core.LAYOUT_ONLY_OK = False


def test_lazy_stat_and_logger():
    from osrl_amd.common.logger import DummyLogger, LazyStat

    class FakeSt:
        keys = ["a"]

        def read_stats(self, step):
            return {"a": 2.0 * step}

    lg = DummyLogger()
    lg.store(a=LazyStat(FakeSt(), 3, "a"))
    lg.store(a=1.0)
    assert lg.get_mean("a") == 3.5 and float(lg.data["a"][0]) == 6.0


def test_lazy_stats_flush_reads_the_ring_once_and_keeps_every_step():
    """store_stats(mode="lazy") over several ring lengths: each flush is ONE read_stats_many call covering the whole
    backlog (keys subset included), every handed-out LazyStat ends with its own step's value, none is lost to a wrap."""
    from osrl_amd.common.logger import DummyLogger, store_stats

    class FakeSt:
        keys = ["a", "b", "c"]
        index = {"a": 0, "b": 1, "c": 2}
        ring_len = 16

        def __init__(self):
            self.host_step, self.calls = 0, 0

        def read_stats_many(self, steps):
            self.calls += 1
            assert all(self.host_step - s < self.ring_len for s in steps)
            return {s: [1.0 * s, 2.0 * s, 3.0 * s] for s in steps}

        def read_stats(self, step=None):
            raise AssertionError("per-row read on the flush path")

    st, lg = FakeSt(), DummyLogger(max_keep=10 ** 6)
    for _ in range(100):
        st.host_step += 1
        store_stats(lg, st, "lazy", keys=["a", "c"])
    assert 100 // 8 - 1 <= st.calls <= 100 // 8 + 1
    done = [v for v in lg.data["c"] if getattr(v, "_val", v) is not None]
    assert len(lg.data["a"]) == 100 and len(done) >= 100 - 8
    assert [float(v) for v in lg.data["c"][:len(done)]] == [3.0 * (i + 1) for i in range(len(done))]


def test_header_is_plain_c():
    """include/osrl_amd.h is the drop-in boundary: it must compile as C99 (no C++ constructs, no torch types) and a
    C translation unit must be able to take the address of every entry point it declares."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "osrl_amd.h")
    names = sorted(set(re.findall(r"\b(osrl_[a-z0-9_]+)\s*\(", open(hdr).read())))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "use.c")
        with open(src, "w") as f:
            f.write('#include "osrl_amd.h"\n#include <stddef.h>\nconst void* osrl_table[] = {\n')
            f.write("".join(f"  (const void*)&{n},\n" for n in names))
            f.write("  NULL};\nint main(void) { return osrl_table[0] == NULL; }\n")
        r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-Wno-pedantic",
                            "-I", os.path.dirname(hdr), "-c", src, "-o", os.path.join(d, "use.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_ctypes_struct_layouts_match_the_compiled_header():
    """sizeof / field offsets of every struct of include/osrl_amd.h as a C compiler sees them == the ctypes mirrors
    in osrl_amd/_lib.py (the binding INTEGRATION.md shows is exactly these classes)."""
    import ctypes as C
    import shutil
    import subprocess
    import tempfile
    from osrl_amd import _lib as L
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    pairs = [("osrl_mlp_t", L.MlpT), ("osrl_pack_entry_t", L.PackEntryT), ("osrl_rows_t", L.RowsT),
             ("osrl_mlp_acts_t", L.ActsT), ("osrl_mlp_grads_t", L.GradsT), ("osrl_mlp_tail_t", L.TailT),
             ("osrl_mlp_seed_t", L.SeedT), ("osrl_dw_entry_t", L.DwEntryT), ("osrl_dw_adam_t", L.DwAdamT), ("osrl_mlp_step_t", L.MlpStepT),
             ("osrl_step_state_t", L.StepStateT), ("osrl_dropout_t", L.DropoutT), ("osrl_env_t", L.EnvT)]
    hdr_dir = os.path.join(ROOT, "include")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "sz.c"), os.path.join(d, "sz")
        with open(src, "w") as f:
            f.write('#include "osrl_amd.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void) {\n')
            for cname, cls in pairs:
                f.write(f'  printf("{cname} %zu", sizeof({cname}));\n')
                for fname, _ in cls._fields_:
                    cfield = {"in_": "in"}.get(fname, fname)  # `in` is a Python keyword: the mirror spells it in_
                    f.write(f'  printf(" %zu", offsetof({cname}, {cfield}));\n')
                f.write('  printf("\\n");\n')
            f.write("  return 0;\n}\n")
        subprocess.run([gcc, "-std=c99", "-I", hdr_dir, src, "-o", exe], check=True, capture_output=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line, (cname, cls) in zip(out, pairs):
        tok = line.split()
        assert tok[0] == cname
        want = [C.sizeof(cls)] + [getattr(cls, fname).offset for fname, _ in cls._fields_]
        assert [int(x) for x in tok[1:]] == want, (cname, tok[1:], want)


def test_plan_rows_are_pinned(monkeypatch):
    """engine/plan.py: the shape-keyed plan chooser returns the pinned rows for the BASELINE configs, and a lab switch in
    the environment changes nothing unless OSRL_LAB=1 says this is a lab run."""
    import warnings
    from osrl_amd.engine import plan as P
    monkeypatch.delenv("OSRL_LAB", raising=False)
    for name, (fn, kw, want) in P.PINNED.items():
        assert fn(**kw) == want, (name, fn(**kw), want)
    monkeypatch.setenv("OSRL_HEAD_TAILS", "0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert P.cpq_plan(76, 2, 2048, 400, 10).head_tails is True       # ignored: not a lab run
    monkeypatch.setenv("OSRL_LAB", "1")
    assert P.cpq_plan(76, 2, 2048, 400, 10).head_tails is False          # a lab run may flip it
    monkeypatch.delenv("OSRL_HEAD_TAILS")
    # the all-CU VAE launches: only the measured region (one round of 48-row tiles, narrow first layer), never without
    # the seeded backward launches, never outside the library's shapes
    assert P.cpq_plan(17, 6, 2048, 400, 10).vae_ns and not P.cpq_plan(17, 6, 2048, 400, 10, seeds=False).vae_ns
    assert not P.cpq_plan(17, 6, 4096, 400, 10).vae_ns and not P.cpq_plan(17, 6, 2048, 256, 10).vae_ns
    assert P.cpq_plan(76, 2, 2048, 400, 10).vae_ns and not P.cpq_plan(17, 6, 512, 400, 10).vae_ns
    # the VAE's Adam goes to the side branch where the action draws ride on the actor launch (C2), not where they are four
    # launches of their own on that branch (C4)
    assert P.cpq_plan(76, 2, 2048, 400, 10).vae_adam_side and not P.cpq_plan(17, 6, 2048, 400, 10).vae_adam_side


def test_every_graph_capture_is_thread_local():
    """A process that holds an RCCL process group has a watchdog thread polling events; in torch's default (global) capture
    mode its queries make an open capture fail and abort the process (DESIGN_LOG round 5).  Every capture of the package
    goes through engine/core.py graph_capture() = thread-local mode -- held here so that a new capture site cannot bring
    the race back."""
    import inspect
    import re
    calls = []
    for dp_, _, files in os.walk(os.path.join(ROOT, "osrl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp_, f)).read()
                calls += [(f, m.group(0)) for m in re.finditer(r"(with|return|=) +torch\.cuda\.graph\([^\n]*", src)]
    assert len(calls) == 1 and calls[0][0] == "core.py" and 'capture_error_mode="thread_local"' in calls[0][1], calls
    from osrl_amd.engine import core
    src = inspect.getsource(core._GraphCapture)
    assert 'capture_error_mode="thread_local"' in src and "_GraphCapture(g)" in inspect.getsource(core.graph_capture)
    # ... and with the cyclic garbage collector off until the capture has ended (round 6: an automatic collection inside a
    # capture destroyed an earlier engine's hipGraph on the capturing thread -- abort()); the switch is put back either way
    assert "gc.disable()" in src and src.count("gc.enable()") >= 2
    import gc
    was = gc.isenabled()

    class _G:  # (no device here: entering the capture raises; the collector must be back on afterwards)
        pass
    try:
        with core.graph_capture(_G()):
            pass
    except BaseException:
        pass
    assert gc.isenabled() == was
