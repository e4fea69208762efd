#!/bin/bash
# usage (GPU box): bash tools/r4_step.sh <tag> -- a development step of round 4: the column-kernel tests, the remap / Riemann timings with
# their probes, the brief bench
TAG=${1:-s}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "remap or riem or edge_profile" > gpurun_out/$TAG/tests.log 2>&1
tail -3 gpurun_out/$TAG/tests.log
(for p in 0 15; do echo probe $p; FV3_MI355X_REMAP_PROBE=$p timeout 200 python tools/remap_time.py 2>&1 | grep -E "^lds|^slabs"; done) > gpurun_out/$TAG/remap_probe.txt 2>&1
cat gpurun_out/$TAG/remap_probe.txt
(RT_FIRST=0 RT_LAST=5 timeout 300 python tools/riem_time.py 2>&1 | grep -E "slab|lds|tolerance") > gpurun_out/$TAG/riem_probe.txt 2>&1
cat gpurun_out/$TAG/riem_probe.txt
bash tools/r4_bench_brief.sh $TAG
