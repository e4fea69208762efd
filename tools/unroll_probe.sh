#!/bin/bash
# grouped row steps of the fused transport kernel (FV3_UNROLL_T): variants built into gpurun_in/*.so against the shipped library
cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/${1:-unroll}; mkdir -p $out
F="--no-cpu --no-model-step --no-cubed --no-general --steps 100 --warmup 20"
for r in 1 2; do
python bench.py $F > $out/base_$r.json 2>/dev/null
for v in gpurun_in/*.so; do n=$(basename $v .so); FV3_MI355X_SO=$PWD/$v python bench.py $F > $out/${n}_$r.json 2>/dev/null; done
done
python - <<P
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); p=b["roofline"]["per_launch"]
        print(f.split('/')[-1], round(b["ms_per_step"],4), b["finite"], {k:round(v["ms_per_step"],3) for k,v in p.items() if k in ("c_sw","d_sw_fused","d_sw_mom_fused")})
    except Exception as e: print(f, "ERR", e)
P
