#!/bin/bash
# usage: tools/lab/build.sh <name> [-D...]   -- builds tools/lab/<name>.hip into tools/lab/bin/<name> (gfx950; runs on the GPU box)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; shift
mkdir -p $R/tools/lab/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans "$@" $R/tools/lab/$N.hip -o $R/tools/lab/bin/$N
