#!/bin/bash
# usage (GPU box): bash tools/r5_ab.sh <tag> [reps] [test-variant] -- A / B of the headline pair: the shipped library against every variants/*.so on
# ONE box; with a third argument, the GPU parity tests of the pair's kernels through that variant first
TAG=${1:-ab}; REPS=${2:-40}; TV=$3
mkdir -p gpurun_out/$TAG
if [ -n "$TV" ]; then
  FV3_MI355X_SO=$PWD/variants/$TV.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c_sw or d_sw or c384 or pair or tp_2d" > gpurun_out/$TAG/tests_$TV.txt 2>&1
  tail -5 gpurun_out/$TAG/tests_$TV.txt
fi
[ -x variants/bw_roof ] && variants/bw_roof > gpurun_out/$TAG/bw_roof.txt 2>&1
SOS=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so
for v in variants/*.so; do [ -f "$v" ] && SOS=$SOS:$PWD/$v; done
FV3_AB_SO=$SOS timeout 900 python tools/pair_ab.py $REPS > gpurun_out/$TAG/pair_ab.txt 2>&1
cat gpurun_out/$TAG/pair_ab.txt
