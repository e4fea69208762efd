#!/bin/bash
# usage (GPU box): bash tools/r3_bench_brief.sh <tag> -- bench.py without the CPU leg, the lines that matter printed
TAG=${1:-x}
python bench.py --no-cpu > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("pair ms", round(d["ms_per_step"], 4), "frac_wall", round(d["roofline"]["pair"]["frac_wall"], 4), "dominant", round(d["roofline"]["frac"], 4))
m = d["model_step"]
print("model sypd", round(m["sypd"], 3), "wall", round(m["wall_s_per_dt_atmos"], 5), "frac", m["whole_step"]["frac"])
print({k: v for k, v in sorted(m["kernels_ms_per_dt_atmos"].items(), key=lambda x: -x[1])[:14]})
c = d["cubed_sphere"]
print("cubed pair", round(c["pair_one_face"]["ms"], 3), "sphere sypd", round(c["sphere_one_gpu"]["sypd"], 4))
print({k: v for k, v in list(c["sphere_one_gpu_kernels_ms_per_dt_atmos"].items())[:16]})
PY
