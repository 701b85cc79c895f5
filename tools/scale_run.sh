#!/bin/bash
# The 1 -> 8 GPU scaling curve of the data-parallel CPQ step (BASELINE.json configs[1] = c2 and configs[3] = c4) on ONE
# node:   bash tools/scale_run.sh [outdir]          (needs as many visible MI355X as the largest N; default 1 2 4 8)
# For every (config, N): `python bench.py --gpus N --config C` (bench.py re-executes itself under torchrun, one rank per
# GPU over RCCL), the JSON line goes to <outdir>/scale_<C>_n<N>.json, and the script checks
#   * rccl_ranks == N for N > 1 (the job really ran on N RCCL ranks),
#   * config.graph -- whether the step incl. its 4 collectives was captured into one hipGraph; if the runtime refused the
#     capture the engine has already fallen back to eager launches on every rank (engine/cpq.py _run + dist.all_agree):
#     the script says so, prints the warning line, and re-runs with OSRL_DP_EAGER=1 so that the eager figure is explicit,
#   * collectives_in_step (N > 1): the four collectives' in-step durations and their share of the step.
# Last it prints value(N) / value(1) per config: the driver computes its own efficiency from the same lines.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}; mkdir -p "$OUT"
NS=${NS:-"1 2 4 8"}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0
for C in c2 c4; do
  for N in $NS; do
    if [ "$N" -gt "$NGPU" ]; then echo "skip $C N=$N: only $NGPU GPU(s) visible"; continue; fi
    F=$OUT/scale_${C}_n${N}.json
    timeout 900 python bench.py --gpus $N --config $C --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $F 2> $OUT/scale_${C}_n${N}.err
    rc=$?
    if [ $rc -ne 0 ] || [ ! -s $F ]; then echo "FAIL $C N=$N rc=$rc"; tail -5 $OUT/scale_${C}_n${N}.err; continue; fi
    python - "$F" "$N" "$OUT/scale_${C}_n${N}.err" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); n = int(sys.argv[2])
assert d["n_gpus"] == n, (d["n_gpus"], n)
if n > 1:
    assert d["rccl_ranks"] == n, f"rccl_ranks {d['rccl_ranks']} != {n}: the job did not run on {n} RCCL ranks"
g = d["config"]["graph"]
line = f"{d['config']['name']} N={n}: {d['value']:.1f} grad-steps/s ({d['optimizer_steps_per_s']:.1f} optimizer steps/s, " \
       f"{d['ms_per_step']:.4f} ms/step), captured graph: {g}, rccl_ranks {d['rccl_ranks']}"
c = d.get("collectives_in_step")
if isinstance(c, list) and c:
    tot = sum(x["us"] for x in c)
    line += f"; collectives in step: " + ", ".join(f"{x['what']} {x['bytes']} B {x['us']:.1f} us" for x in c) + \
            f" = {tot:.1f} us = {tot / (10 * d['ms_per_step']):.1f} % of the step"
print(line)
if n > 1 and not g:
    w = [l for l in open(sys.argv[3], errors='replace') if 'capture' in l.lower()]
    print("  capture REFUSED -> eager collectives on every rank; runtime said:", (w[-1].strip() if w else "(no warning line found)"))
    sys.exit(7)
PY
    if [ $? -eq 7 ]; then
      OSRL_DP_EAGER=1 timeout 900 python bench.py --gpus $N --config $C --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $OUT/scale_${C}_n${N}_eager.json 2>> $OUT/scale_${C}_n${N}.err
      python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f\"  eager: {d['value']:.1f} grad-steps/s\")" $OUT/scale_${C}_n${N}_eager.json
    fi
  done
  python - "$OUT" "$C" <<'PY'
import glob, json, os, sys
out, c = sys.argv[1], sys.argv[2]
v = {}
for f in glob.glob(os.path.join(out, f"scale_{c}_n*.json")):
    if f.endswith("_eager.json"):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    v[d["n_gpus"]] = d["value"]
if 1 in v:
    print(f"{c} scaling vs N=1: " + ", ".join(f"N={n}: {v[n] / v[1]:.2f}x" for n in sorted(v)))
PY
done
