#!/bin/bash
# usage (on the GPU box): bash tools/refresh_profiles.sh v13 -- the runs behind profiles/r01_<tag>_*: the -m gpu suite, bench.py,
# the config-5-sized step, tracers per wavefront, rocprofv3 --kernel-trace --stats of bench.py, SQ counters of the model step.
# Everything lands in gpurun_out/<tag>/; copy what is to be kept into profiles/.
TAG=${1:-vX}
mkdir -p gpurun_out/$TAG
R=$PWD
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/$TAG/pytest.txt
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python tools/bench_config5.py > gpurun_out/$TAG/config5.json 2> gpurun_out/$TAG/config5.err
python tools/bench_tracer.py > gpurun_out/$TAG/tracer_nt.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu > /tmp/kt.log 2>&1
cd $R
DB=$(find /tmp/kt -name "*_results.db" | head -1)
python profiles/summarize_rocprof.py r01_$TAG $DB > gpurun_out/$TAG/summarize.log 2>&1
cp profiles/r01_$TAG* gpurun_out/$TAG/ 2>/dev/null
tail -3 /tmp/kt.log >> gpurun_out/$TAG/summarize.log
bash tools/pmc_model_step.sh $TAG
