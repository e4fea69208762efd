mkdir -p gpurun_out/ab14
B=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so; V=$PWD/variants
PAIR_NO_HEAT=1 FV3_AB_SO=$B:$V/bf12.so:$V/bf12_m3.so:$V/bf12.so@MARCH_TJ_MOM=55:$V/bf12_m3.so@MARCH_TJ_MOM=64:$V/bf12_m3.so@MARCH_TJ_MOM=39 timeout 1500 python tools/pair_ab2.py 5 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab14/pair_ab2.txt
FV3_MI355X_SO=$V/bf12.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c_sw or d_sw or c384 or pair or tp_2d or sponge or golden or mixed" 2>&1 | tail -3
