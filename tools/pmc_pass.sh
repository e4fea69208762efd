#!/bin/bash
# counters of the cubed-sphere pass kernels of one face pair (tools/bench_cubed.py --nh)
R=$PWD
mkdir -p gpurun_out/pmc_pass
cd /tmp && export TMPDIR=/tmp
n=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pp$n -- python $R/tools/bench_cubed.py --nh --steps 3 > /tmp/pp$n.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ppk -- python $R/tools/bench_cubed.py --nh --steps 3 > /tmp/ppk.log 2>&1
cd $R
python - > gpurun_out/pmc_pass/pmc_pass.csv <<'PY'
import glob, sqlite3
print("kernel,counter,avg_per_launch,launches")
for db in sorted(glob.glob("/tmp/pp[0-9]/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        rows = list(con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        rows = [("ERR " + str(e), "", 0, 0)]
    for k, c, v, m in rows:
        if "Pass" in k or "A2B" in k or "Tp2dFrame" in k or "ERR" in k:
            print(f"\"{k.replace('fv3::','')[:90]}\",{c},{v:.1f},{m}")
for db in glob.glob("/tmp/ppk/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    for n_, c_, t_, a_, p_ in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "Pass" in n_ or "A2B" in n_ or "Tp2dFrame" in n_:
            print(f"\"{n_.replace('fv3::','')[:90]}\",avg_us,{a_:.2f},{c_}")
PY
tail -3 /tmp/pp5.log
