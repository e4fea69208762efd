#!/bin/bash
# usage (on the GPU box): bash tools/r6_round.sh <tag> -- the round's evidence in one call: HBM counters of the pair and of the remap (bench.py
# reports them), the bench line, kernel-trace stats of the same command, and (round 5) the SQ counters of the pair's kernels:
# instructions issued, wavefront / wait cycles -> gpurun_out/<tag>/pmc_sq_pair.csv
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
bash tools/pmc_hbm_pair.sh $TAG
bash tools/pmc_remap.sh $TAG
bash tools/pmc_riem.sh $TAG      # round 6: the Riemann solvers' counters every round (none were taken in round 5)
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
# SQ counters of the pair (one --pmc group per pass, with --kernel-trace only)
cd /tmp
n=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  n=$((n+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/psq$n -- python $R/tools/run_pair.py 3 > /tmp/psq$n.log 2>&1
done
cd $R
python - > gpurun_out/$TAG/pmc_sq_pair.csv <<'PY'
import glob, sqlite3
print("kernel,counter,avg_per_launch,launches")
for db in sorted(glob.glob("/tmp/psq*/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        rows = list(con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception:
        rows = []
    for k, c, v, m in rows:
        if "March" in k or "Fused" in k:
            kk = k.split("fv3::")[2].split(">")[0] + ">" if k.count("fv3::") > 1 else k[:60]
            print(f"\"{kk}\",{c},{v:.1f},{m}")
PY
