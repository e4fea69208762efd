#!/bin/bash
# usage (GPU box): bash tools/r4_lanes.sh <tag> -- the two lanes of the cubed-sphere pair: bit comparison against the one-lane order
# (C96 L32 both tiles, C384 L127), the cubed-sphere tests, the pair's timing with and without, the brief bench
TAG=${1:-l}
mkdir -p gpurun_out/$TAG
(NPX=97 NPZ=32 REPS=8 timeout 600 python tools/lanes_check.py; NPX=97 NPZ=32 REPS=4 NH=0 timeout 600 python tools/lanes_check.py; NPX=385 NPZ=127 REPS=3 timeout 900 python tools/lanes_check.py) > gpurun_out/$TAG/lanes_check.txt 2>&1
grep -E "tile|lanes_check|DIFF|Error|error" gpurun_out/$TAG/lanes_check.txt | head -30
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "cubed or face_group or sphere" > gpurun_out/$TAG/tests.log 2>&1
tail -3 gpurun_out/$TAG/tests.log
(python tools/bench_cubed.py --nh; FV3_MI355X_SIDE_STREAM=0 python tools/bench_cubed.py --nh; python tools/bench_cubed.py --nh --prod) > gpurun_out/$TAG/bench_cubed.txt 2>&1
python - gpurun_out/$TAG/bench_cubed.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("pair_ms", round(d["pair_ms"], 3), {k: v for k, v in d["per_label"].items() if v[1] > 0.08})
PY
bash tools/r4_bench_brief.sh $TAG
