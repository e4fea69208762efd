#!/bin/bash
# usage: tools/bench_launches.sh [label]   -- one short bench.py run, prints ms_per_step and the per-launch averages
python bench.py --no-cpu --no-model-step | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['roofline']['launches'].items()})"
