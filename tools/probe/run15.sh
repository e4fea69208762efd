mkdir -p gpurun_out/r05_loop
for nx in 384 512; do
  FV3_BENCH_LOOPBACK=1 timeout 600 python bench.py --nx $nx --no-cpu --no-model-step --no-cubed --no-general --steps 40 > gpurun_out/r05_loop/loopback_$nx.json 2> gpurun_out/r05_loop/loopback_$nx.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r05_loop/loopback_$nx.json').read().strip().splitlines()[-1])
print($nx, 'loopback ms', d['ms_per_step'], d['config']['halo'], {k:round(v['ms_per_step'],4) for k,v in d['roofline']['per_launch'].items()})
"
done
timeout 2000 python -m pytest tests -x -q -m gpu -k "fortran or restarted or rccl_loopback" 2>&1 | tail -4
