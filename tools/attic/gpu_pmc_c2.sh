#!/bin/bash
# PMC counters of the CDT step's kernels (C5, eager launches so that every dispatch is a counted kernel): one --pmc pass
# per counter group with --kernel-trace only
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_c2; rm -rf $O; mkdir -p $O
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
G3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
for g in 1 2 3; do
  eval C=\$G$g
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -f csv -d $O/g$g -o p -- python $GRAFT_REPO_ROOT/bench.py --config c2 --eager --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-roofline --no-cold > $O/g$g.log 2>&1)
done
python - <<'PY'
import csv, glob, json, os, re
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_c2"
acc = {}
for f in glob.glob(f"{O}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void\s+", "", r.get("Kernel_Name", "")).replace("(anonymous namespace)::", "")
        k = re.split(r"[(]", k)[0][:60]
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["n"] = max(len(v) for v in cs.values())
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        m["kernel_cycles"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: m["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
        if "SQ_INSTS_VALU" in m: m["valu_inst_per_simd_cycle"] = m["SQ_INSTS_VALU"] / (cyc * 1024)
        if "SQ_INSTS_MFMA" in m: m["mfma_inst_x32_per_simd_cycle"] = m["SQ_INSTS_MFMA"] * 32 / (cyc * 1024)
    out[k] = m
json.dump(out, open(f"{O}/pmc_c2.json", "w"), indent=1)
for k, m in sorted(out.items(), key=lambda kv: -kv[1].get("kernel_cycles", 0) * kv[1].get("n", 0))[:26]:
    print(k[:40].ljust(40), {a: (round(b, 3) if b < 100 else int(b)) for a, b in m.items() if a in ("n", "kernel_cycles", "mfma_util", "valu_inst_per_simd_cycle", "mfma_inst_x32_per_simd_cycle", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVES")})
PY
rm -rf $O/g1 $O/g2 $O/g3
