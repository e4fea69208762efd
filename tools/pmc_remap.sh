#!/bin/bash
# usage (on the GPU box): bash tools/pmc_remap.sh <tag>
# The remap alone (tools/remap_time.py, C384L127 tile, 4 tracers), slab kernels and LDS kernels: HBM traffic (FETCH_SIZE doubled -- gfx950
# correction, profiles/README.md -- and WRITE_SIZE, separate passes) and the SQ counters that say what a kernel waits for
# -> gpurun_out/<tag>/pmc_remap.csv (kernel, counter, average per launch)
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
n=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  n=$((n+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/pr$n -- python $R/tools/remap_time.py > /tmp/pr$n.log 2>&1
done
cd $R
python - > gpurun_out/$TAG/pmc_remap.csv <<'PY'
import glob, sqlite3
print("kernel,counter,avg_per_launch")
for db in sorted(glob.glob("/tmp/pr*/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        rows = list(con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        rows = []
    for k, c, v in rows:
        if "emap" in k:
            print(f"\"{k[:60]}\",{c},{v:.1f}")
PY
python - > gpurun_out/$TAG/hbm_traffic_remap.json <<'PY'
import glob, json, sqlite3, sys
sys.path.insert(0, ".")
from gfdl_atmos_cubed_sphere_amd import lib
def pmc(n, name):
    dbs = glob.glob(f"/tmp/pr{n}/**/*_results.db", recursive=True)
    con = sqlite3.connect(dbs[0])
    return {k: v for k, v in con.execute("select kernel_name, avg(value) from counters_collection where counter_name=? group by kernel_name", (name,))}
f, w = pmc(1, "FETCH_SIZE"), pmc(2, "WRITE_SIZE")
def total(pats):
    return sum(v * 1024 * 2 for k, v in f.items() if any(p in k for p in pats)) + sum(v * 1024 for k, v in w.items() if any(p in k for p in pats))
print(json.dumps({"build_id": lib.build_id(), "shape": [384, 384, 127, 4],
                  "_note": "HBM bytes of one Lagrangian_to_Eulerian call (every remap_* launch) on a 384x384x127 tile with 4 tracers: rocprofv3 --pmc "
                           "FETCH_SIZE (KB, doubled: gfx950 correction) + WRITE_SIZE (KB), separate passes, tools/pmc_remap.sh; algorithmic: "
                           "(144 + 16 nq) B per cell = 3.90e9",
                  "slabs": total(["RemapCoords", "RemapFields", "RemapDelzFinal", "RemapPe"]),
                  "lds": total(["RemapFastScalars", "RemapFastWind", "RemapPe"])}, indent=1))
PY
tail -n 3 /tmp/pr1.log /tmp/pr3.log > gpurun_out/$TAG/pmc_remap.log 2>&1
