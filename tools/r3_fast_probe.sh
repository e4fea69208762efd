#!/bin/bash
# round 3: the fast column solvers on the GPU -- parity tests, then bench legs with and without FV3_MI355X_FAST=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -k "fast" 2>&1 | tail -15 > gpurun_out/r03_fast_tests.log
python bench.py --no-cubed --no-cpu --no-general --steps 20 > gpurun_out/r03_fast_off.json 2> gpurun_out/r03_fast_off.err
FV3_MI355X_FAST=1 python bench.py --no-cubed --no-cpu --no-general --steps 20 > gpurun_out/r03_fast_on.json 2> gpurun_out/r03_fast_on.err
python - <<'PY'
import json
for n in ("off","on"):
    try:
        d=json.load(open(f"gpurun_out/r03_fast_{n}.json")); m=d["model_step"]
        print(n, "pair ms", round(d["ms_per_step"],3), "sypd", round(m["sypd"],3), "whole", m["whole_step"]["frac"], {k:v["ms_per_call"] for k,v in m["column_kernels"].items()})
    except Exception as e: print(n, "ERR", e)
PY
cat gpurun_out/r03_fast_tests.log
