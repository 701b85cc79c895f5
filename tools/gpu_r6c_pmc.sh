#!/bin/bash
# round 6 (third session): HBM traffic (PMC) of the N*B-row launches in the forms the step now ships -- encoder on shared-observation tiles,
# target cost critics on the selected row list -- separate --pmc passes with --kernel-trace only -> profiles/pmc_traffic.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6cpmc; rm -rf $O; mkdir -p $O
for cfg in c2 c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -f csv -d $O/${cfg}_$c -o p -- python $GRAFT_REPO_ROOT/tools/pmc_nb.py $cfg 20 > $O/${cfg}_$c.log 2>&1)
    tail -n 1 $O/${cfg}_$c.log
  done
done
python - <<'PY'
import csv, glob, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6cpmc"
names = {"mlp_fwd_nb8_kernel": "mlp_fwd<vae-encoder, N*B rows>", "mlp_fwd_nb_kernel": "mlp_fwd<cost_critic_old x2, N*B rows>"}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/pmc_nb.py = the "
                 "engine's own N*B-row launches in isolation IN THE FORMS THE STEP SHIPS (round 6, third session: the encoder on "
                 "shared-observation tiles; the target cost critics on the row list of the step's select launch, a quarter of "
                 "the N*B rows), 20 launches each, mean per launch, counters in KB; traffic = 2 x FETCH_SIZE + WRITE_SIZE: "
                 "calibrated in round 6 on known-byte streams (tools/fetch_calib.hip, profiles/r6_fetch_calib.json: FETCH_SIZE "
                 "reports 0.500 of the bytes at 4, 8 AND 16 bytes per lane, WRITE_SIZE 1.000) -- tools/gpu_r6c_pmc.sh; static, "
                 "not measured in the bench run"}
raw = {}
for cfg in ("c2", "c4"):
    out[cfg] = {}
    for sym, label in names.items():
        tot, ok = 0.0, True
        for c, w in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            vals = []
            for f in glob.glob(f"{O}/{cfg}_{c}/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r.get("Kernel_Name", "")
                    stem = sym[:-len("_kernel")]  # (the shared-tile forms are ..._pre_kernel)
                    hit = stem in k and not (stem == "mlp_fwd_nb" and "nb8" in k)
                    if hit and r.get("Counter_Name") == c:
                        vals.append(float(r["Counter_Value"]))
            raw[f"{cfg}/{sym}/{c}"] = {"n": len(vals), "mean_kb": sum(vals) / max(len(vals), 1)}
            ok = ok and len(vals) > 0
            tot += w * sum(vals) / max(len(vals), 1)
        out[cfg][label] = tot * 1024 if ok else None
out["raw_kb"] = raw
json.dump(out, open(f"{O}/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k in ("c2", "c4")}, indent=1)); print(json.dumps(raw))
PY
