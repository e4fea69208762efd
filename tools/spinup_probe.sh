#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/${1:-spinup}; mkdir -p $out
F="--no-cpu --no-model-step --no-cubed --no-general --steps 20 --warmup 5"
for r in 1 2 3; do
for sp in 0 100 400 1500; do
  FV3_BENCH_SPINUP=$sp python bench.py $F > $out/sp${sp}_$r.json 2>/dev/null
  sleep 5
done; done
python - <<P
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(b["ms_per_step"],4))
P
