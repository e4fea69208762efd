mkdir -p gpurun_out/ab6
export FV3_MI355X_SO=$PWD/variants/bf5.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c_sw or d_sw or c384 or pair or tp_2d or sponge" > gpurun_out/ab6/tests.txt 2>&1; tail -4 gpurun_out/ab6/tests.txt
unset FV3_MI355X_SO
FV3_AB_SO=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so:$PWD/variants/bf5.so timeout 600 python tools/pair_ab.py 40 2>&1 | grep -v amdgpu.ids > gpurun_out/ab6/pair_ab.txt; cat gpurun_out/ab6/pair_ab.txt
FV3_AB_SO=$PWD/variants/bf5.so timeout 900 python tools/probe/tj_sweep.py "" "CSW=16" "CSW=32" "CSW=48" "CSW=64" "CSW=96" "FUSED=32" "FUSED=48" "FUSED=64" "FUSED=77" "FUSED=96" "MOM=32" "MOM=48" "MOM=64" "MOM=96" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/ab6/tj.txt; cat gpurun_out/ab6/tj.txt
