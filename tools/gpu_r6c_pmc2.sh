#!/bin/bash
# round 6 (third session): the per-kernel counter table of the C2 and C4 steps under the final plan (plan.ood_rows: select / row-set
# forward / sum / Polyak launches; eager launches: every dispatch a counted kernel) -> profiles/r6c_pmc_c2.json / r6c_pmc_c4.json.
# Counters in passes of their own with --kernel-trace only (MI355X_MICROARCH.md).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6cpmc2; rm -rf $O; mkdir -p $O
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for cfg in c2 c4; do
for g in 1 2; do
  eval C=\$G$g
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -f csv -d $O/${cfg}_g$g -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --eager --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-roofline --no-cold > $O/${cfg}_g$g.log 2>&1)
done
CFG=$cfg python - <<'PY'
import csv, glob, json, os, re
cfg = os.environ["CFG"]
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6cpmc2"
acc = {}
for f in glob.glob(f"{O}/{cfg}_g*/**/*counter_collection.csv", recursive=True):
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
json.dump(out, open(f"{O}/r6c_pmc_{cfg}.json", "w"), indent=1)
print(cfg)
for k, m in sorted(out.items(), key=lambda kv: -kv[1].get("kernel_cycles", 0) * kv[1].get("n", 0))[:22]:
    print(" ", k[:44].ljust(44), {a: (round(b, 3) if b < 100 else int(b)) for a, b in m.items() if a in ("n", "kernel_cycles", "mfma_util", "valu_inst_per_simd_cycle", "mfma_inst_x32_per_simd_cycle", "SQ_WAVES")})
PY
done
rm -rf $O/c2_* $O/c4_* 2>/dev/null; ls $O
