#!/bin/bash
# usage (on the GPU box): bash tools/r2_round.sh <tag> -- bench line, kernel-trace stats of the same command, HBM counters of the pair
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
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
bash tools/pmc_hbm_pair.sh $TAG
