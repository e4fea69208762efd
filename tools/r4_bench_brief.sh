#!/bin/bash
# usage (GPU box): bash tools/r4_bench_brief.sh <tag> [bench args] -- bench.py without the CPU leg, the lines that matter printed
TAG=${1:-x}; shift
python bench.py --no-cpu "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("pair ms", round(d["ms_per_step"], 4), "frac_wall", round(d["roofline"]["pair"]["frac_wall"], 4), "dominant", round(d["roofline"]["frac"], 4))
m = d["model_step"]
if m and "error" not in m:
    print("model sypd (", m["column_solvers"][:14], ")", round(m["sypd"], 3), "wall", round(m["wall_s_per_dt_atmos"], 5), "frac", m["whole_step"]["frac"])
    t = m.get("tolerance_mode")
    if t:
        print("  tolerance mode sypd", round(t["sypd"], 3), "fast_vs_parity", {k: float("%.2e" % v) for k, v in t["fast_vs_parity"].items()})
    print({k: v for k, v in sorted(m["kernels_ms_per_dt_atmos"].items(), key=lambda x: -x[1])[:14]})
    print({k: (v["ms_per_call"], v["frac"]) for k, v in m["column_kernels"].items()})
else:
    print("model_step", m)
c = d["cubed_sphere"]
if c and "error" not in c:
    print("cubed pair", round(c["pair_one_face"]["ms"], 3), "sphere sypd", round(c["sphere_one_gpu"]["sypd"], 4), c["sphere_one_gpu"].get("face_group"))
    print(c["sphere_one_gpu"]["kernel_breakdown"])
    print({k: v for k, v in list(c["sphere_one_gpu_kernels_ms_per_dt_atmos"].items())[:16]})
    c2 = c.get("config2_c96_l79_hydrostatic")
    print("config2", {k: c2.get(k) for k in ("sypd", "wall_s_per_dt_atmos", "face_group", "error", "kernel_breakdown")})
else:
    print("cubed", c)
PY
