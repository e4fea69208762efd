#!/bin/bash
# usage (GPU box): bash tools/r4_lanes3.sh <tag> -- production namelist (every level damped): the momentum half beside the transport half;

TAG=${1:-l}
mkdir -p gpurun_out/$TAG
(echo "PROD NH"; PROD=1 NPX=385 NPZ=127 REPS=3 timeout 900 python tools/lanes_check.py; echo "PROD hydrostatic"; PROD=1 NH=0 NPX=385 NPZ=127 REPS=2 timeout 900 python tools/lanes_check.py; echo "PROD NH C96 L32"; PROD=1 NPX=97 NPZ=32 REPS=6 timeout 900 python tools/lanes_check.py) > gpurun_out/$TAG/lanes_check.txt 2>&1
grep -E "PROD|tile|lanes_check|DIFF|Error|error" gpurun_out/$TAG/lanes_check.txt | head -40
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "two_lanes or damping or production" > gpurun_out/$TAG/tests.log 2>&1
grep -E "passed|failed" gpurun_out/$TAG/tests.log | tail -2
