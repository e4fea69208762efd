mkdir -p gpurun_out/ab12
B=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so; V=$PWD/variants
FV3_AB_SO=$B:$V/bf9.so:$V/bf10.so:$V/bf10.so@SPONGE_MARCH=0:$V/bf10.so@MARCH_TJ_FUSED=48,MARCH_TJ_MOM=48:$V/bf10.so@MARCH_TJ_FUSED=64,MARCH_TJ_MOM=64:$V/bf10.so@MARCH_TJ_CSW=32:$V/bf10.so@MARCH_TJ_CSW=48 timeout 1500 python tools/pair_ab2.py 5 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab12/pair_ab2.txt
echo "== heat NULL"; FV3_AB_SO=$V/bf10.so PAIR_NO_HEAT=1 timeout 600 python tools/pair_ab2.py 4 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab12/pair_ab2_noheat.txt
FV3_MI355X_DEBUG_SEGMENTS=1 FV3_AB_SO=$V/bf10.so timeout 300 python tools/pair_ab2.py 1 1 2>&1 | grep segments | sort -u
