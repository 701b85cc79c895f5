"""Build libosrl_amd.so (hipcc, gfx950 only) IN-TREE at osrl_amd/lib/.

The .so is git-ignored but travels to the GPU box with the source snapshot.
``python -m osrl_amd.build`` or ``osrl_amd.build.build()``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libosrl_amd.so")
SOURCES = ["mlp.hip", "mlp_nb.hip", "mlp_nb64.hip", "mlp_dw.hip", "vae_ns.hip", "optim.hip", "rng.hip", "glue.hip", "cdt.hip", "env.hip", "ingest.hip", "bear.hip", "dice.hip", "act.hip", "ipc.hip", "diag.hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libosrl_amd.so for gfx950)")


HEADERS = ["philox.h", "step.h", "argmem.h", "adam.h", "gather.h", "mlp_common.h", "dwt.h", "gelu.h", "trace.h"]  # csrc headers shared between translation units


def _common_deps():
    """What every object depends on: the public header, the shared csrc headers, this file (the compiler flags)."""
    return [os.path.join(PKG, "..", "include", "osrl_amd.h")] + [os.path.join(CSRC, h) for h in HEADERS] + [__file__]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _common_deps()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # one builder at a time (torchrun starts N ranks that all import the package): the others wait on the lock and
    # then find the library fresh
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         # SimplifyCFG's store sinking merges "ring[i] = load" of sibling branches into a store through a
         # pointer phi, after which the weight-ring arrays of mlp.hip can no longer be promoted to registers
         # (they end up in scratch with an s_waitcnt vmcnt(0) right after the prefetch loads)
         "-mllvm", "-sink-common-insts=false", "-Wno-pass-failed",
         # kernarg preload (gfx940+): the command processor hands the first 16 dwords of a launch's arguments to every
         # wave in SGPRs, instead of each wave issuing s_loads from the kernarg segment.  Where the runtime keeps
         # kernel arguments in host memory those s_loads are PCIe round trips at the head of every wave: CPQ step
         # 1884 -> 2092 steps/s there, 2176 -> 2190 with device-resident kernargs (profiles/r3_kernarg_ab.txt)
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]
# per-file additions.  cdt.hip: its only MFMA users are the attention kernels, whose accumulators are consumed by vector
# instructions right away (mask, softmax, dS) -- results in AGPRs cost one v_accvgpr_read per element (192 in the
# backward's listing); with the VGPR form the register file is one pool (160 -> 131 registers) and the reads are gone
# (tools/attn_lab.hip: backward 263.6 -> 258.8 us at C5's shape)
FILE_FLAGS = {"cdt.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
OPTIONAL_FLAGS = {("-mllvm", "-amdgpu-mfma-vgpr-form")}  # speed only: dropped where the compiler does not know them
OBJDIR = os.path.join(LIBDIR, "obj")  # git-ignored (*.o); the objects are a build cache only
_flag_ok: dict = {}


def _flag_supported(hip: str, flag: tuple) -> bool:
    """An unknown ``-mllvm`` option is FATAL to hipcc: probe an optional one once on an empty translation unit
    and drop it where the compiler predates it -- the kernels are correct without it."""
    if flag not in _flag_ok:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as f:
                f.write("__global__ void k() {}\n")
            r = subprocess.run([hip, "--offload-arch=gfx950", *flag, "-c", src, "-o", os.path.join(d, "probe.o")],
                               capture_output=True)
            _flag_ok[flag] = r.returncode == 0
    return _flag_ok[flag]


def file_flags(hip: str, source: str) -> list:
    out, fl = [], FILE_FLAGS.get(source, [])
    i = 0
    while i < len(fl):
        pair = tuple(fl[i:i + 2]) if fl[i] == "-mllvm" else (fl[i],)
        if pair not in OPTIONAL_FLAGS or _flag_supported(hip, pair):
            out += list(pair)
        i += len(pair)
    return out


def _flag_key(cmd_flags: list) -> str:
    import hashlib
    return hashlib.sha256(" ".join(cmd_flags).encode()).hexdigest()[:16]


def _build_locked(verbose: bool, force: bool = False) -> str:
    """One object per translation unit (compiled in parallel, re-compiled only when its source, philox.h or the
    header is newer), then one link."""
    os.makedirs(OBJDIR, exist_ok=True)
    hip = _hipcc()
    common = _common_deps()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s.replace(".hip", ".o"))
        extra = [os.path.join(CSRC, "mlp_nb.hip")] if s == "mlp_nb64.hip" else []  # (it is that file, compiled again)
        flags = FLAGS + file_flags(hip, s)
        # the flag set is part of an object's staleness key (a changed flag must rebuild it: mtimes alone would not)
        keyf, key = obj + ".flags", _flag_key(flags)
        try:
            same_flags = open(keyf).read().strip() == key
        except OSError:
            same_flags = False
        stale = force or not os.path.exists(obj) or not same_flags or any(
            os.path.getmtime(d) > os.path.getmtime(obj) for d in [src] + extra + common)
        if stale:
            cmd = [hip] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((s, subprocess.Popen(cmd), keyf, key))
    failed = []
    for s, p, keyf, key in jobs:
        if p.wait() != 0:
            failed.append(s)
        else:
            with open(keyf, "w") as f:
                f.write(key + "\n")
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [hip, "--offload-arch=gfx950", "-shared", "-fPIC"] + \
        [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES] + \
        ["-L" + os.path.join(os.path.dirname(os.path.dirname(hip)), "lib"), "-lhsa-runtime64", "-o", tmp]  # diag.hip
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
