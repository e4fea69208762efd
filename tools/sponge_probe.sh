#!/bin/bash
# what the two sponge levels (LDS-tile kernels on the side stream) cost the pair: headline / no sponge / serialised
cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/${1:-sponge}; mkdir -p $out
F="--no-cpu --no-model-step --no-cubed --no-general --steps 100 --warmup 20"
for r in 1 2; do
python bench.py $F > $out/base_$r.json 2> $out/base_$r.err
FV3_BENCH_SPONGE=0 python bench.py $F > $out/nosponge_$r.json 2> $out/nosponge_$r.err
FV3_MI355X_SIDE_STREAM=0 python bench.py $F > $out/serial_$r.json 2> $out/serial_$r.err
done
python - <<P
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); p=b["roofline"]["per_launch"]
        print(f, round(b["ms_per_step"],4), {k:round(v["ms_per_step"],3) for k,v in p.items()})
    except Exception as e: print(f, "ERR", e)
P
