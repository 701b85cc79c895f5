#!/bin/bash
# HBM traffic of the N*B forward kernels by PMC (separate --pmc passes, no trace flags besides kernel-trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc; rm -rf $O; mkdir -p $O
for k in "enc 80" "q2 80"; do set -- $k
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace -f csv -d $O/${1}_$c -o p -- python $GRAFT_REPO_ROOT/tools/kone.py $1 20480 $2 20 > $O/${1}_$c.log 2>&1)
  done
done
python - <<'PY'
import csv, glob, json, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc"
out={}
for name in ("enc","q2"):
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        fs=glob.glob(f"{O}/{name}_{c}/**/*counter_collection.csv", recursive=True)
        vals=[]
        for f in fs:
            for r in csv.DictReader(open(f)):
                if "mlp_fwd" in r.get("Kernel_Name","") and r.get("Counter_Name")==c:
                    vals.append(float(r["Counter_Value"]))
        out[f"{name}_{c}"]={"n":len(vals),"mean":sum(vals)/max(len(vals),1), "min":min(vals) if vals else None, "max": max(vals) if vals else None}
json.dump(out, open(f"{O}/pmc_raw.json","w"), indent=1)
print(json.dumps(out, indent=1))
# bytes per launch = FETCH_SIZE + WRITE_SIZE (rocprofv3 reports both in KB on this build: x 1024).  gfx950 correction
# (MI355X_MICROARCH.md, HBM section): FETCH_SIZE halves WIDE (16 B / lane) streaming reads; this kernel's HBM-side
# reads are the 4-byte-per-lane stage-in of its input rows (the 16 B / lane weight fragments are L2 hits), which the
# counter reports at face value -- calibrated on the kernel itself: 6.4 MB of input + 0.8 MB of weights fetched once
# = 7.2 MB expected, 6.6 MB counted (part of the weights stays in the Infinity Cache between launches)
traffic = {"mlp_fwd<vae-encoder, N*B rows>": (out["enc_FETCH_SIZE"]["mean"] + out["enc_WRITE_SIZE"]["mean"]) * 1024,
           "mlp_fwd<cost_critic_old x2, N*B rows>": (out["q2_FETCH_SIZE"]["mean"] + out["q2_WRITE_SIZE"]["mean"]) * 1024}
json.dump(traffic, open(f"{O}/pmc_traffic.json","w"), indent=1)
print(json.dumps(traffic, indent=1))
PY
