#!/bin/bash
# c_sw on a cubed-sphere face (CswMarch<KPW, 0, true>): levels per wavefront x rows per segment
cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/${1:-cswcubed}; mkdir -p $out
run() { python tools/bench_cubed.py --nh --steps 20 2>/dev/null | tail -1; }
echo "base $(run)" > $out/log.txt
for kpw in 1 2 3; do for tj in 24 32 48 64 96 128; do
  echo "kpw$kpw tj$tj $(FV3_MI355X_CSW_KPW=$kpw FV3_MI355X_MARCH_TJ_CSW=$tj run)" >> $out/log.txt
done; done
python - <<P
import json
for l in open("$out/log.txt"):
    tag, js = l.split(" {",1)[0], "{"+l.split(" {",1)[1]
    d=json.loads(js); print(tag, round(d["pair_ms"],3), d["per_label"].get("c_sw"), d["per_label"].get("d_sw_fused"), d["per_label"].get("d_sw_mom_fused"))
P
