#!/bin/bash
# usage (on the GPU box): bash tools/pmc_hbm_model_step.sh v13 -- HBM-side traffic of every kernel of the model step:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (together they do not fit the TCC counter slots), durations from
# a --kernel-trace --stats run of the same command.  FETCH_SIZE is doubled (gfx950 correction, see profiles/README.md).
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu --steps 2 --warmup 1"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -- $CMD > /tmp/pf.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -- $CMD > /tmp/pw.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pk -- $CMD > /tmp/pk.log 2>&1
cd $R
python - $(find /tmp/pf -name "*_results.db" | head -1) $(find /tmp/pw -name "*_results.db" | head -1) $(find /tmp/pk -name "*_results.db" | head -1) > gpurun_out/$TAG/pmc_hbm_model_step.csv <<'PY'
import sqlite3, sys
def pmc(db, name):
    con = sqlite3.connect(db)
    return {k: (v, n) for k, v, n in con.execute(
        "select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (name,))}
f, w = pmc(sys.argv[1], "FETCH_SIZE"), pmc(sys.argv[2], "WRITE_SIZE")
dur = {n: a for n, a in sqlite3.connect(sys.argv[3]).execute("select name, average from top_kernels")}
print("kernel,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,fetch_bytes_corrected(x2),write_bytes,total_bytes,avg_duration_us(kernel-trace run),TB_per_s")
rows = []
for k in f:
    fk, n = f[k]
    wk = w.get(k, (0., 0))[0]
    fb, wb = fk * 1024 * 2, wk * 1024
    us = dur.get(k, 0.) / 1e3 if dur.get(k, 0.) > 1e4 else dur.get(k, 0.)
    rows.append((fb + wb, '"%s",%d,%.1f,%.1f,%.4g,%.4g,%.4g,%.1f,%.2f' % (k, n, fk, wk, fb, wb, fb + wb, us, (fb + wb) / (us * 1e-6) / 1e12 if us else 0.)))
for _, r in sorted(rows, reverse=True):
    print(r)
PY
tail -2 /tmp/pf.log /tmp/pw.log >> gpurun_out/$TAG/pmc.log
