#!/bin/bash
# usage: tools/build_variant.sh <name> [patch-script.py | -D...]...   -- builds variants/<name>.so from a COPY of csrc/ (+ include/) under /tmp,
# optionally patched by python scripts (run inside the copy's csrc dir) and / or with extra -D flags.  The product sources are not touched.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/fv3_variant_$NAME
rm -rf $W; mkdir -p $W/pkg/csrc $W/include $R/variants
cp $R/gfdl_atmos_cubed_sphere_amd/csrc/*.h $R/gfdl_atmos_cubed_sphere_amd/csrc/*.hip $W/pkg/csrc/
cp $R/include/*.h $W/include/
DEFS=""
for a in "$@"; do
  case "$a" in
    -D*) DEFS="$DEFS $a";;
    *) (cd $W/pkg/csrc && python "$R/$a");;
  esac
done
cd $W/pkg/csrc
( time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fPIC -shared $DEFS fv3_api.hip -o $R/variants/$NAME.so ) > $W/build.log 2>&1
tail -3 $W/build.log
