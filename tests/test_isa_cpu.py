"""Compile-time guards for properties of the HIP kernels that no numeric test sees (CPU only: hipcc cross-compiles
gfx950 assembly without a GPU).

Round 2 found most single-workgroup kernels of the CPQ step paying one memory round trip per load: the compiler sinks a
load into the branch that consumes it and turns ``c ? x[i] : 0`` back into such a branch ("last
stretch").  The sources now request their loads together (clamped address + select / OR, ``pin()``); these tests hold
the kernels to the resulting number of wait groups, and the N*B-row forward to its register / LDS-store shape, so that a
compiler or source change that undoes it fails here."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
pytestmark = pytest.mark.skipif(HIPCC is None, reason="no hipcc")


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    from osrl_amd.build import FILE_FLAGS, FLAGS
    from isa_loads import scan
    d = tmp_path_factory.mktemp("isa")
    procs = {}
    for name in ("glue", "optim", "mlp", "mlp_nb", "vae_ns", "cdt"):
        out = str(d / f"{name}.s")
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(f"{name}.hip", []) + \
            ["-S", "--cuda-device-only", os.path.join(ROOT, "osrl_amd", "csrc", f"{name}.hip"), "-o", out]
        procs[name] = (subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), out)
    res = {}
    for name, (p, out) in procs.items():
        assert p.wait() == 0, f"hipcc -S failed on {name}.hip"
        res[name] = scan(out)
    return res


def _one(table, *needles):
    hits = [k for k in table if all(n in k for n in needles)]
    assert len(hits) == 1, (needles, hits)
    return table[hits[0]]


@pytest.mark.parametrize("needles,max_waits", [
    (("quantile_kernel",), 18),                  # 73 loads; was one wait per load (17 with compile-time launch geometry)
    (("cpq_ood_stat_kernel", "ILb1E"), 14),      # 98 loads; was 88 wait groups (12 before the geometry became constants)
    (("cpq_critic_loss_kernelILb1E",), 3),       # was 8 per batch row
    (("cpq_critic_loss_kernel_pILb1E",), 4),     # its device-resident-argument twin (+ the descriptor's own loads)
    (("cpq_cost_loss_kernelILb1E",), 6),
    (("cpq_cost_loss_kernel_pILb1E",), 7),
    (("cpq_actor_loss_kernel", "ILb1E"), 2),
    (("vae_loss_kernelE",), 3),
    (("vae_loss_kernel_pE",), 4),
    (("vae_latent_bwd_kernel",), 1),
    (("vae_kl_rows_kernel",), 2),
    (("gauss_head_bwd_kernel",), 3),
])
def test_chain_kernels_request_their_loads_together(listings, needles, max_waits):
    r = _one(listings["glue"], *needles)
    assert r["waits"] <= max_waits, (needles, r)
    assert r["scratch"] == 0, (needles, r)


def test_adam_requests_every_operand_up_front(listings):
    for needle in ("adam_kernelE", "adam_kernel_pE"):  # by-value kernel and its device-resident-descriptor twin
        r = _one(listings["optim"], needle)
        assert r["loads"] >= 14 and r["waits"] <= 4 and r["scratch"] == 0, (needle, r)  # slabs, p, m, v, target, two maps: one group


def test_nb_forward_kernel_shape(listings):
    """mlp_fwd_nb_kernel: no scratch, one wave's registers within the file, float4 epilogue stores."""
    # (third template argument, round 6: the LIST instantiation that reads a device-chosen row set -- <4, false, true>;
    # fourth, second session: tiles of shared src0 rows (osrl_rows_t.share0) -- an instantiation of its own so that the plain
    # form keeps its registers: the chain kernels of the other graph branch fit beside <= 208 of them)
    regs = {}
    for inst in ("ILi7ELb1ELb0ELb0E", "ILi4ELb0ELb0ELb0E", "ILi7ELb0ELb0ELb0E", "ILi4ELb0ELb1ELb0E", "ILi4ELb0ELb0ELb1E"):
        shapes = []
        # by-value kernel and its device-resident-descriptor twin (csrc/argmem.h): the same body behind one extra
        # scalar load -- same registers, same stores
        for needles in (("mlp_fwd_nb_kernelI", inst), ("mlp_fwd_nb_kernel_pI", inst)):
            r = _one(listings["mlp_nb"], *needles)
            assert r["scratch"] == 0 and 0 < r["vgprs"] <= 512, (needles, r)
            assert r["b128_writes"] >= 60, (needles, r)  # the transposed-tile epilogue (ds_write_b32 per element before)
            shapes.append((r["vgprs"], r["b128_writes"]))
        assert shapes[0] == shapes[1], (inst, shapes)
        regs[inst] = shapes[0][0]
    assert regs["ILi4ELb0ELb0ELb0E"] <= 208 and regs["ILi4ELb0ELb0ELb1E"] <= 208, regs
    for needle in ("mlp_fwd_nb8_kernelE", "mlp_fwd_nb8_kernel_pE", "mlp_fwd_nb8_pre_kernelE", "mlp_fwd_nb8_pre_kernel_pE"):
        r = _one(listings["mlp_nb"], needle)  # the 8-wave 25-block form and its shared-row twin: two <= 160-register waves per
        assert r["scratch"] == 0 and 0 < r["vgprs"] <= 160, (needle, r)  # SIMD leave room for two 96-register chain waves


def test_descriptor_pointer_kernels_read_their_arguments_with_scalar_loads(listings):
    """The "_p" kernels (descriptor in device memory, reached through the constant address space) must not fall back
    to per-lane loads of the descriptor: same VGPR count and no scratch as their by-value twins."""
    for twin in (("mlp_fwd_kernelI", "mlp_fwd_kernel_pI"), ("mlp_fwd2_kernelI", "mlp_fwd2_kernel_pI"),
                 ("mlp_bwd_dz_kernelI", "mlp_bwd_dz_kernel_pI")):
        for inst in ("Li1ELi2ELi8E", "Li1ELi4ELi8E"):
            a, b = _one(listings["mlp"], twin[0], inst), _one(listings["mlp"], twin[1], inst)
            assert b["scratch"] == 0 and a["scratch"] == 0, (twin, inst, a, b)
            assert abs(a["vgprs"] - b["vgprs"]) <= 4, (twin, inst, a["vgprs"], b["vgprs"])


def test_one_launch_step_kernel_fits_its_registers(listings):
    """mlp_step_kernel (the BC step as one launch): 8 waves per workgroup = 2 per SIMD = 256 registers per lane for the
    forward / backward bodies, the 64 x 64 (or 32 x 32 with all four k-steps in flight) dW tile and the preloaded
    optimizer state together -- no scratch in the captured ("_p") instances, and both twins the same shape."""
    for inst in ("ILi2E", "ILi4E"):
        a, b = _one(listings["mlp"], "mlp_step_kernelI", inst), _one(listings["mlp"], "mlp_step_kernel_pI", inst)
        # (the by-value twin -- eager mode and the capture's warm-up pass -- indexes the gather descriptor's small arrays
        # dynamically out of the kernarg segment, which the compiler stages through <= 64 bytes of scratch)
        assert b["scratch"] == 0 and a["scratch"] <= 64, (inst, a, b)
        for r in (a, b):
            assert 0 < r["vgprs"] <= 256, (inst, r)
        assert abs(a["vgprs"] - b["vgprs"]) <= 16, (inst, a["vgprs"], b["vgprs"])


def test_vae_ns_kernels_fit_beside_the_nb_launch(listings):
    """csrc/vae_ns.hip: the wide launches run beside the N*B-row launch of the other graph branch (one 197-register wave
    per SIMD, 84 KB of LDS), so one wave per SIMD of theirs has to fit in 512 - 200 = 312 registers -- a kernel that
    grows past it does not fail, it WAITS for CUs to retire (round 3 measured 12 us of work taking 84-88 us that way).
    Held here for the instances the CPQ plan launches (the <*, 2> forms belong to 4096-row / wide-latent shapes that the
    auto rule does not pick); no scratch anywhere; the descriptor-in-memory twins the same shape."""
    t = listings["vae_ns"]
    for needle in ("fwd_enc_kernelILi1E", "gen_kernelILi0ELi1E", "gen_kernelILi1ELi1E", "gen_kernelILi2ELi1E",
                   "gen_kernelILi2ELi2E"):
        a, b = _one(t, "vae_ns_" + needle), _one(t, "vae_ns_" + needle.replace("kernelI", "kernel_pI"))
        for r in (a, b):
            assert r["scratch"] == 0 and 0 < r["vgprs"] <= 312, (needle, r)
    for k, r in t.items():
        assert r["scratch"] == 0, (k, r)


def test_attention_kernels_keep_their_residency(listings):
    """csrc/cdt.hip, round 5's attention kernels (built with the file's own flags, build.py FILE_FLAGS: MFMA results in
    VGPRs): the backward for 5 row blocks on 3 waves runs FOUR workgroups per CU = 3 waves per SIMD (<= 168 registers; its
    33 KB of LDS allow the same four), the forward six (<= 84 ... 128 registers keeps four); no scratch in any
    instantiation, and no accumulator-register traffic (`v_accvgpr_*`) left in the listing's register file split."""
    t = listings["cdt"]
    bwd = _one(t, "attn_bwd_v_kernelILi5ELi2ELi3E")
    fwd = _one(t, "attn_fwd_v_kernelILi5ELi2ELi3E")
    assert bwd["scratch"] == 0 and 0 < bwd["vgprs"] <= 168, bwd
    assert fwd["scratch"] == 0 and 0 < fwd["vgprs"] <= 128, fwd
    n = 0
    for k, r in t.items():
        if "attn_fwd_v_kernel" in k or "attn_bwd_v_kernel" in k:
            n += 1
            assert r["scratch"] == 0 and r["vgprs"] <= 256, (k, r)
    assert n == 16, n  # 2 kernels x (5 | 8 row blocks) x (head width 16 | 32) x 2 wave counts
