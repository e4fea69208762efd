#!/bin/bash
# usage (on the GPU box): bash tools/pmc_hbm_pair.sh <tag>
# HBM-side traffic of the kernels of the c_sw + d_sw pair at 384 x 384 x 127 for BOTH geometry modes, stamped with the build id
# of the sources (gfdl_atmos_cubed_sphere_amd.lib.build_id) -> gpurun_out/<tag>/hbm_traffic.json; copy it to
# profiles/hbm_traffic.json: bench.py reports roofline.traffic from it only while the build id matches.
# FETCH_SIZE and WRITE_SIZE are collected in SEPARATE rocprofv3 passes (together they do not fit the TCC counter slots);
# FETCH_SIZE is doubled (gfx950 correction, profiles/README.md).
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for GEOM in 2 0; do
  if [ $GEOM = 0 ]; then export FV3_MI355X_GEOM=0; else unset FV3_MI355X_GEOM; fi
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf$GEOM -- python $R/tools/run_pair.py 3 > /tmp/pf$GEOM.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw$GEOM -- python $R/tools/run_pair.py 3 > /tmp/pw$GEOM.log 2>&1
done
unset FV3_MI355X_GEOM
cd $R
python - > gpurun_out/$TAG/hbm_traffic.json <<'PY'
import glob, json, sqlite3, sys
sys.path.insert(0, ".")
from gfdl_atmos_cubed_sphere_amd import lib
LABEL = {"CswMarch": "c_sw", "DswTransportFused": "d_sw_fused", "DswMomentumFused": "d_sw_mom_fused"}
def pmc(pattern, name):
    dbs = glob.glob(pattern, recursive=True)
    if not dbs:
        return {}
    con = sqlite3.connect(dbs[0])
    return {k: v for k, v in con.execute(
        "select kernel_name, avg(value) from counters_collection where counter_name=? group by kernel_name", (name,))}
out = {"build_id": lib.build_id(),
       "_note": "bytes per launch at 384x384x127: rocprofv3 --pmc FETCH_SIZE (KB, doubled: gfx950 correction of MI355X_MICROARCH.md, "
                "calibrated in profiles/r01_v8_pmc_hbm.csv) + WRITE_SIZE (KB), separate passes, tools/pmc_hbm_pair.sh. Infinity-Cache hits "
                "are counted; the L2 atomics' read side (cx, cy, mfx, mfy in d_sw_fused, ~0.6 GB) is not part of FETCH_SIZE."}
for geom in (2, 0):
    f = pmc(f"/tmp/pf{geom}/**/*_results.db", "FETCH_SIZE")
    w = pmc(f"/tmp/pw{geom}/**/*_results.db", "WRITE_SIZE")
    for k, fk in f.items():
        for pat, lab in LABEL.items():
            if pat in k:
                out[f"{lab}@geom{geom}"] = fk * 1024 * 2 + w.get(k, 0.0) * 1024
                out[f"{lab}@geom{geom}_read_write"] = [fk * 1024 * 2, w.get(k, 0.0) * 1024]
print(json.dumps(out, indent=1))
PY
tail -2 /tmp/pf2.log /tmp/pw2.log /tmp/pf0.log /tmp/pw0.log > gpurun_out/$TAG/pmc_hbm_pair.log 2>&1
