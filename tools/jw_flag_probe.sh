#!/bin/bash
# which damping settings run the JW wave stably (tools/jw_long_run.py --flags)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python tools/jw_long_run.py --days 1.5 --flags "$1" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); s=d['series']
print(sys.argv[1], sys.argv[2], [ (x['day'], round(x['ps_min_hPa'],2), round(x['u_max'],1)) for x in s[1:] ])" "$1" "$2"; }
run '{"nord":2,"d4_bg":0.02}' tiny
run '{"nord":2,"d4_bg":0.0}' zero
FV3_MI355X_MARCH=0 run '{"nord":1}' passes_only
run '{"nord":0,"d2_bg":0.02}' nord0
