for P in 0 256 512 768 1024 2048; do
  FV3_MI355X_COL_POOL=$P python bench.py --no-cubed --no-general --no-cpu --steps 20 --warmup 10 > gpurun_out/pool_$P.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/pool_$P.json'));m=d['model_step'];k=m['kernels_ms_per_dt_atmos'];print($P, m['sypd'], {n:k[n] for n in ('riem_solver3','riem_solver_c','remap_fields')})"
done
