#!/bin/bash
# usage (on the GPU box): bash tools/r3_round.sh <tag> -- the round's evidence in one call: bench line, kernel-trace stats of the same
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
