#!/bin/bash
# usage (on the GPU box): bash tools/r4_round.sh <tag> -- the round's evidence in one call: bench line, kernel-trace stats of the same
# command, HBM counters of the pair (both geometry modes) and of the remap (parity and fast kernels), SQ counters of the remap
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
# the counter traffic first: bench.py reports roofline.traffic / column_kernels.remap.traffic from profiles/hbm_traffic*.json of THIS build
bash tools/pmc_hbm_pair.sh $TAG
bash tools/pmc_remap.sh $TAG
cp gpurun_out/$TAG/hbm_traffic.json profiles/hbm_traffic.json
cp gpurun_out/$TAG/hbm_traffic_remap.json profiles/hbm_traffic_remap.json
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu --no-model-step > /tmp/kt.log 2>&1
cd $R
DB=$(find /tmp/kt -name "*_results.db" | head -1)
python - $DB > gpurun_out/$TAG/kernel_stats.csv <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
print("name,calls,total_us,avg_us,percent")
for n, c, t, a, p in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"\"{n}\",{c},{t:.3f},{a:.3f},{p:.2f}")
PY
# round 4: the exact Riemann kernels' SQ counters, the timing probes
bash tools/pmc_riem.sh $TAG > gpurun_out/$TAG/pmc_riem.log 2>&1
(for p in 0 1 2 4 7; do echo probe $p; FV3_MI355X_RIEM_PROBE=$p RT_FIRST=1 RT_LAST=2 timeout 200 python tools/riem_time.py 2>&1 | grep lds; done; RT_FIRST=0 RT_LAST=5 timeout 300 python tools/riem_time.py 2>&1 | grep -E "slab|lds|tolerance") > gpurun_out/$TAG/riem_probe.txt 2>&1
(for p in 0 1 2 4 8 15; do echo probe $p; FV3_MI355X_REMAP_PROBE=$p timeout 200 python tools/remap_time.py 2>&1 | grep -E "^lds|^slabs"; done) > gpurun_out/$TAG/remap_probe.txt 2>&1
# round 4, second half: the two lanes of the cubed-sphere pair against the one-lane order (bit comparison + wall time, C384 L127 NH / hydrostatic)
(NPX=385 NPZ=127 REPS=3 timeout 900 python tools/lanes_check.py; FV3_MI355X_LANE_D2=1 NPX=385 NPZ=127 REPS=2 timeout 900 python tools/lanes_check.py; NH=0 NPX=385 NPZ=127 REPS=2 timeout 900 python tools/lanes_check.py) 2>&1 | grep -E "tile|lanes_check|DIFF" > gpurun_out/$TAG/lanes_check.txt
