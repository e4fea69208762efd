#!/bin/bash
# usage (on the GPU box): bash tools/lab/pmc.sh <binary> <out.csv> [args...]   with LAB_ONLY=<label text> in the environment
# SQ counters of one lab variant (separate --pmc passes, --kernel-trace only) -> per-kernel averages
B=$1; OUT=$2; shift; shift
mkdir -p $(dirname $OUT)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/labpmc*
n=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/labpmc$n -- $R/$B "$@" > /tmp/labpmc$n.log 2>&1
done
cd $R
python3 - > $OUT <<'PY'
import glob, sqlite3
print("kernel,counter,avg_per_launch,launches")
for db in sorted(glob.glob("/tmp/labpmc*/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        rows = list(con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        rows = []
    for k, c, v, m in rows:
        print(f"\"{k[:90]}\",{c},{v:.1f},{m}")
PY
