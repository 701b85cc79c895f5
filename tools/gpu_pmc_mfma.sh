#!/bin/bash
# MFMA-busy / wave-state counters of the dominant kernel (VAE encoder on the 20480 sampled rows), one --pmc pass per
# group with --kernel-trace only (no other trace domain), for the 8-wave and the 4-wave form of the 80-row kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma; rm -rf $O; mkdir -p $O
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for W in 8 4; do for g in 1 2; do
  eval C=\$G$g
  (cd /tmp && OSRL_NB_WAVES=$W rocprofv3 --pmc $C --kernel-trace -f csv -d $O/w${W}_g$g -o p -- python $GRAFT_REPO_ROOT/tools/kone.py enc 20480 80 20 > $O/w${W}_g$g.log 2>&1)
done; done
python - <<'PY'
import csv, glob, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_mfma"
out = {}
for W in (8, 4):
    acc = {}
    for g in (1, 2):
        for f in glob.glob(f"{O}/w{W}_g{g}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "mlp_fwd_nb" in r.get("Kernel_Name", ""):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    m["n_dispatches"] = {k: len(v) for k, v in acc.items()}
    # gfx94x derived-counter formula (MI355X_MICROARCH.md: rocprofv3 ships no gfx950 section): MfmaUtil =
    # SQ_VALU_MFMA_BUSY_CYCLES / (GPU cycles * CUs * 4 SIMDs).  rocprofv3 SUMS a counter over the 8 XCDs: GRBM_GUI_ACTIVE
    # comes back as 8 x the kernel's cycles (1.5 M for a 71 us launch at 2.4 GHz = 8 x 188 k), the SQ counters as the
    # total over all CUs.  SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_16x16x4_f32 (3.968 M MFMAs -> 127 M).
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        m["kernel_cycles"] = cyc
        m["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m and "GRBM_GUI_ACTIVE" in m:
        # MOPS counts 512-FLOP units (gfx94x formula); fp32 MFMA peak = 256 FLOP/clk/CU
        m["mfma_flop_per_clk_per_cu"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256)
        m["mfma_flop_frac_of_peak"] = m["mfma_flop_per_clk_per_cu"] / 256.0
    out[f"nb_waves_{W}"] = m
json.dump(out, open(f"{O}/pmc_mfma.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
