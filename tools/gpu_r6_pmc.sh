#!/bin/bash
# round 6: HBM traffic (PMC) of the SHIPPED N*B-row kernels, C2 and C4 -- separate --pmc passes with --kernel-trace only
# (MI355X_MICROARCH.md, HBM section) -> profiles/pmc_traffic.json (keyed by config; bench.py's roofline.traffic);
# then the per-kernel counter table of the C2 and C4 steps (eager launches: every dispatch a counted kernel) ->
# profiles/r6_pmc_c2.json / r6_pmc_c4.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6pmc; rm -rf $O; mkdir -p $O
# FETCH_SIZE / WRITE_SIZE against known-byte streams at 4 / 8 / 16 bytes per lane (tools/fetch_calib.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace -f csv -d $O/calib_$c -o p -- /tmp/fetch_calib > $O/calib_$c.log 2>&1)
done
python - <<'PY'
import csv, glob, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6pmc"
known = float(1 << 30)
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{O}/calib_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                res.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
out = {"known_bytes_per_launch": known, "what": "tools/fetch_calib.hip: 1 GiB read once at 4 / 8 / 16 bytes per lane (read_w<1|2|4>), 1 GiB written at 4 bytes per lane (write_1); counter values in KB; ratio = counter * 1024 / known bytes"}
for k, cs in sorted(res.items()):
    out[k] = {c: {"mean_kb": sum(v) / len(v), "n": len(v), "ratio_to_known": sum(v) / len(v) * 1024 / known} for c, v in cs.items()}
json.dump(out, open(f"{O}/r6_fetch_calib.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
for cfg in c2 c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -f csv -d $O/${cfg}_$c -o p -- python $GRAFT_REPO_ROOT/tools/pmc_nb.py $cfg 20 > $O/${cfg}_$c.log 2>&1)
  done
done
python - <<'PY'
import csv, glob, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6pmc"
names = {"mlp_fwd_nb8_kernel": "mlp_fwd<vae-encoder, N*B rows>", "mlp_fwd_nb_kernel": "mlp_fwd<cost_critic_old x2, N*B rows>"}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/pmc_nb.py = the "
                 "engine's own N*B-row launches in isolation, 20 launches each, mean per launch; counters in KB on this build "
                 "(x 1024); FETCH_SIZE taken at face value: these kernels' HBM-side reads are 4-byte-per-lane row staging, the "
                 "16-byte weight fragments are L2 hits (the guide's x2 correction applies to wide streaming reads only) -- "
                 "tools/gpu_r6_pmc.sh, round 6; static, not measured in the bench run"}
raw = {}
for cfg in ("c2", "c4"):
    out[cfg] = {}
    for sym, label in names.items():
        tot = 0.0
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            vals = []
            for f in glob.glob(f"{O}/{cfg}_{c}/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r.get("Kernel_Name", "")
                    hit = (sym + "_p" in k or sym + "(" in k or sym + "<" in k) and not (sym == "mlp_fwd_nb_kernel" and "nb8" in k)
                    if hit and r.get("Counter_Name") == c:
                        vals.append(float(r["Counter_Value"]))
            raw[f"{cfg}/{sym}/{c}"] = {"n": len(vals), "mean_kb": sum(vals) / max(len(vals), 1)}
            tot += sum(vals) / max(len(vals), 1)
        out[cfg][label] = tot * 1024 if tot > 0 else None
json.dump(out, open(f"{O}/pmc_traffic.json", "w"), indent=1)
json.dump(raw, open(f"{O}/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1)); print(json.dumps(raw, indent=1))
PY
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
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6pmc"
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
json.dump(out, open(f"{O}/r6_pmc_{cfg}.json", "w"), indent=1)
print(cfg)
for k, m in sorted(out.items(), key=lambda kv: -kv[1].get("kernel_cycles", 0) * kv[1].get("n", 0))[:22]:
    print(" ", k[:44].ljust(44), {a: (round(b, 3) if b < 100 else int(b)) for a, b in m.items() if a in ("n", "kernel_cycles", "mfma_util", "valu_inst_per_simd_cycle", "mfma_inst_x32_per_simd_cycle", "SQ_WAVES")})
PY
done
rm -rf $O/c2_* $O/c4_* $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE 2>/dev/null; ls $O
