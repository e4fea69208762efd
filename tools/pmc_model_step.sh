#!/bin/bash
# usage (on the GPU box): bash tools/pmc_model_step.sh v13 -- SQ counters (one pass) of every kernel of the model step
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pm -- python $R/bench.py --no-cpu --steps 2 --warmup 1 > /tmp/pm.log 2>&1
cd $R
DB=$(find /tmp/pm -name "*_results.db" | head -1)
python - "$DB" > gpurun_out/$TAG/pmc_sq_model_step.csv <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"))
d = {}
for k, c, v, n in rows:
    d.setdefault(k, {})[c] = v
    d[k]["n"] = n
cs = ["SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "GRBM_GUI_ACTIVE"]
print("kernel,dispatches," + ",".join(cs) + ",waves_per_simd,valu_issue_frac")
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["n"]):
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print('"%s",%d,' % (k, v["n"]) + ",".join("%.6g" % v.get(c, 0) for c in cs) + ",%.2f,%.3f" % (v.get("SQ_WAVE_CYCLES", 0) * 4 / g / 1024, v.get("SQ_INSTS_VALU", 0) * 4 / 1024 / g))
PY
tail -2 /tmp/pm.log >> gpurun_out/$TAG/pmc.log
