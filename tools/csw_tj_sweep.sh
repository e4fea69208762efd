#!/bin/bash
# c_sw rows-per-segment sweep on the headline tile (uniform-metric kernel, default 24): FV3_MI355X_MARCH_TJ_CSW
cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/${1:-cswtj}; mkdir -p $out
F="--no-cpu --no-model-step --no-cubed --no-general --steps 100 --warmup 20"
for r in 1 2; do
for tj in 0 16 20 24 28 32 48 64 96; do
  if [ $tj = 0 ]; then python bench.py $F > $out/base_$r.json 2>/dev/null
  else FV3_MI355X_MARCH_TJ_CSW=$tj python bench.py $F > $out/tj${tj}_$r.json 2>/dev/null; fi
done; done
python - <<P
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); p=b["roofline"]["per_launch"]
        print(f.split('/')[-1], round(b["ms_per_step"],4), {k:round(v["ms_per_step"],3) for k,v in p.items() if k in ("c_sw","d_sw_fused","d_sw_mom_fused")})
    except Exception as e: print(f, "ERR", e)
P
