mkdir -p gpurun_out/ab9
B=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so; V=$PWD/variants
S="SPONGE_MARCH=0,ROUND_SIMDS=0"
FV3_AB_SO=$B:$V/bf8.so:$V/bf8.so@SPONGE_MARCH=0:$V/bf8.so@ROUND_SIMDS=0:$V/bf8.so@$S:$V/v_cur_atomics.so@$S:$V/v_cur_branchy.so@$S:$V/v_transport_general.so@ROUND_SIMDS=0:$V/v_transport_general.so timeout 1500 python tools/pair_ab2.py 5 20 2>&1 | grep -v amdgpu.ids > gpurun_out/ab9/pair_ab2.txt; cat gpurun_out/ab9/pair_ab2.txt
