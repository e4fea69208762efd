V=$PWD/variants
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['per_label']
print('$1', 'pair', round(d['pair_ms'],3), 'c_sw', p['c_sw'][1], 'fused', p['d_sw_fused'][1], 'mom', p['d_sw_mom_fused'][1], 'passes', round(sum(v[1] for k,v in p.items() if k not in ('c_sw','d_sw_fused','d_sw_mom_fused')),3))
"; }
for r in 1 2; do
python tools/bench_cubed.py --nh 2>/dev/null | show base
FV3_MI355X_CSW_KPW=1 python tools/bench_cubed.py --nh 2>/dev/null | show kpw1
FV3_MI355X_SO=$V/csw2w.so python tools/bench_cubed.py --nh 2>/dev/null | show csw2w
FV3_MI355X_SO=$V/csw2w.so FV3_MI355X_CSW_KPW=1 python tools/bench_cubed.py --nh 2>/dev/null | show csw2w_kpw1
FV3_MI355X_MARCH_TJ_CSW=32 python tools/bench_cubed.py --nh 2>/dev/null | show tjcsw32
FV3_MI355X_MARCH_TJ_FUSED=55 python tools/bench_cubed.py --nh 2>/dev/null | show tjf55
done
