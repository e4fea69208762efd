mkdir -p gpurun_out/ab11
B=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so; V=$PWD/variants
echo "== default levels"; FV3_AB_SO=$V/bf9.so:$V/bf9.so@SPONGE_MARCH=0:$V/bf9.so@ROUND_SIMDS=0 timeout 600 python tools/pair_ab2.py 4 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab11/a.txt
echo "== no sponge levels (127 plain)"; PAIR_NSPONGE=-1 FV3_AB_SO=$V/bf9.so:$V/bf9.so@SPONGE_MARCH=0:$V/bf9.so@ROUND_SIMDS=0 timeout 600 python tools/pair_ab2.py 4 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab11/b.txt
echo "== npz 125, no sponge"; NPZ=125 PAIR_NSPONGE=-1 FV3_AB_SO=$V/bf9.so:$V/bf9.so@ROUND_SIMDS=0 timeout 600 python tools/pair_ab2.py 4 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab11/c.txt
echo "== npz 129, no sponge"; NPZ=129 PAIR_NSPONGE=-1 FV3_AB_SO=$V/bf9.so:$V/bf9.so@ROUND_SIMDS=0 timeout 600 python tools/pair_ab2.py 4 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab11/d.txt
