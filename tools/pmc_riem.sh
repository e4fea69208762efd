#!/bin/bash
# usage (on the GPU box): bash tools/pmc_riem.sh <tag>
# The two Riemann solvers alone (tools/riem_time.py, C384L127 tile), parity and fast kernels: HBM traffic (FETCH_SIZE doubled -- gfx950
# correction, profiles/README.md -- and WRITE_SIZE, separate passes) and the SQ counters that say what a kernel waits for
# -> gpurun_out/<tag>/pmc_riem.csv (kernel, counter, average per launch)
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
n=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  n=$((n+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/pq$n -- python $R/tools/riem_time.py > /tmp/pq$n.log 2>&1
done
cd $R
python - > gpurun_out/$TAG/pmc_riem.csv <<'PY'
import glob, sqlite3
print("kernel,counter,avg_per_launch")
for db in sorted(glob.glob("/tmp/pq*/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        rows = list(con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        rows = []
    for k, c, v in rows:
        if "iem" in k:
            print(f"\"{k[:70]}\",{c},{v:.1f}")
PY
