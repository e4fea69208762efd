mkdir -p gpurun_out/ab13
V=$PWD/variants
L=""
for t in 36 40 43 48 52; do L="$L:$V/bf11.so@MARCH_TJ_FUSED=$t,MARCH_TJ_MOM=$t"; done
PAIR_NO_HEAT=1 FV3_AB_SO=$V/bf11.so$L:$V/bf11.so@MARCH_TJ_CSW=20:$V/bf11.so@MARCH_TJ_CSW=28:$V/bf11.so@MARCH_TJ_CSW=36 timeout 1500 python tools/pair_ab2.py 5 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab13/pair_ab2.txt
