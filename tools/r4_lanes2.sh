#!/bin/bash
# usage (GPU box): bash tools/r4_lanes2.sh <tag> -- the two lanes on one C384 L127 face: with Courant numbers of the frame's own
# (FV3_MI355X_LANE_D2=1, the default) and without, the bit comparison against the one-lane order, the two-lane tests, the brief bench
TAG=${1:-l}
mkdir -p gpurun_out/$TAG
(echo "LANE_D2=1"; NPX=385 NPZ=127 REPS=3 timeout 900 python tools/lanes_check.py; echo "LANE_D2=0"; FV3_MI355X_LANE_D2=0 NPX=385 NPZ=127 REPS=3 timeout 900 python tools/lanes_check.py; echo "hydrostatic"; NH=0 NPX=385 NPZ=127 REPS=3 timeout 900 python tools/lanes_check.py; echo "C768 L79 (config 5's face)"; NH=0 NPX=769 NPZ=79 REPS=2 timeout 900 python tools/lanes_check.py) > gpurun_out/$TAG/lanes_check.txt 2>&1
grep -E "LANE|hydro|C768|tile|lanes_check|DIFF|Error|error" gpurun_out/$TAG/lanes_check.txt | head -40
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "two_lanes or cubed_c384 or cubed_c48 or cubed_hybrid or face_group or hip_graph" > gpurun_out/$TAG/tests.log 2>&1
grep -E "passed|failed" gpurun_out/$TAG/tests.log | tail -2
bash tools/r4_bench_brief.sh $TAG
