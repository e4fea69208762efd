#!/bin/bash
# round 3: the JW wave with weaker / higher-order divergence damping on the corrected corner area_c
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { python tools/jw_long_run.py --days "$3" --flags "$1" 2>gpurun_out/jw_probe_err.log | python -c "
import json,sys
d=json.load(sys.stdin); s=d['series']
print(sys.argv[1], sys.argv[2], [ (x['day'], round(x['ps_min_hPa'],2), round(x['u_max'],1)) for x in s[1:] ])" "$1" "$2"; }
{
run '{"nord":2,"d4_bg":0.12}' n2_012 1.5
run '{"nord":2,"d4_bg":0.15}' n2_015 1.5
run '{"nord":3,"d4_bg":0.12}' n3_012 1.5
run '{"nord":3,"d4_bg":0.15}' n3_015 1.5
run '{"nord":1,"d4_bg":0.05}' n1_005 1.5
run '{"nord":3,"d4_bg":0.15,"do_vort_damp":true,"vtdm4":0.03}' n3_015_vd 10
run '{"nord":3,"d4_bg":0.15,"do_vort_damp":true,"vtdm4":0.03,"d_con":1.0}' n3_015_vd_dcon 10
} > gpurun_out/r03_jw_nord_probe.txt 2>&1
cat gpurun_out/r03_jw_nord_probe.txt
