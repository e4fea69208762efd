python -m pytest tests -m gpu -x -q -k "remap or fv_dyn or fv_cycle or fortran" 2>&1 | tail -3
for scr in 0 1; do
  echo "== REMAP_SCR=$scr"
  FV3_MI355X_REMAP_SCR=$scr python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['model_step']; k=m['kernels_ms_per_dt_atmos']
print('wall', round(m['wall_s_per_dt_atmos']*1e3,2), {x:k[x] for x in ('remap_fields','remap_coords','remap_delz_final','riem_solver3')})"
  FV3_MI355X_REMAP_SCR=$scr python tools/bench_config5.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_dt_atmos']
print('config5 wall', round(d['wall_s_per_dt_atmos']*1e3,2), d['finite'], {x:k[x] for x in ('remap_fields','tracer_step','remap_coords','remap_delz_final')})"
done
