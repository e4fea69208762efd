mkdir -p gpurun_out/v13
R=$PWD
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/v13/pytest.txt
python bench.py > gpurun_out/v13/bench.json 2> gpurun_out/v13/bench.err
python tools/bench_config5.py > gpurun_out/v13/config5.json 2> gpurun_out/v13/config5.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu > /tmp/kt.log 2>&1
cd $R
DB=$(find /tmp/kt -name "*_results.db" | head -1)
python profiles/summarize_rocprof.py r01_v13 $DB > gpurun_out/v13/summarize.log 2>&1
cp profiles/r01_v13* gpurun_out/v13/ 2>/dev/null
tail -3 /tmp/kt.log >> gpurun_out/v13/summarize.log
