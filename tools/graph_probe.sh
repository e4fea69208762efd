#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/${1:-graph}; mkdir -p $out
F="--no-cpu --no-model-step --no-cubed --steps 60 --warmup 10"
for r in 1 2 3; do
FV3_BENCH_PAIR_GRAPH=0 python bench.py $F > $out/eager_$r.json 2> $out/eager_$r.err
python bench.py $F > $out/graph_$r.json 2> $out/graph_$r.err
done
python - <<P
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(b["ms_per_step"],4), b["finite"], b["config"]["launch"], round(b["general_metrics"]["ms_per_step"],4), b["roofline"]["pair"]["frac_wall"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
P
