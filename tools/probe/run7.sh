mkdir -p gpurun_out/ab7
SOS=$PWD/gfdl_atmos_cubed_sphere_amd/csrc/libfv3_mi355x.so
for v in variants/*.so; do [ -f "$v" ] && SOS=$SOS:$PWD/$v; done
FV3_MI355X_DEBUG_SEGMENTS=1 FV3_AB_SO=$SOS timeout 900 python tools/pair_ab.py 40 2>&1 | grep -v amdgpu.ids | sort -u -k1,1 -k2,2 --stable > /dev/null
FV3_AB_SO=$SOS timeout 900 python tools/pair_ab.py 40 2>&1 | grep -v amdgpu.ids > gpurun_out/ab7/pair_ab.txt; cat gpurun_out/ab7/pair_ab.txt
FV3_MI355X_DEBUG_SEGMENTS=1 FV3_AB_SO=$PWD/variants/bf7_m3.so timeout 300 python tools/probe/tj_sweep.py "" 2>&1 | grep "segments" | sort -u > gpurun_out/ab7/segments.txt; cat gpurun_out/ab7/segments.txt
