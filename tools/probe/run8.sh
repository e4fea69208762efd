mkdir -p gpurun_out/ab8
SOS=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so
for v in variants/bf5.so variants/bf6.so variants/bf7.so variants/bf7_m3.so; do [ -f "$v" ] && SOS=$SOS:$PWD/$v; done
FV3_AB_SO=$SOS timeout 1200 python tools/pair_ab2.py 6 20 2>&1 | grep -v amdgpu.ids > gpurun_out/ab8/pair_ab2.txt; cat gpurun_out/ab8/pair_ab2.txt
