#!/bin/bash
# Tuning only: a side build of libdfe with extra macros for ONE translation unit (default: the cost-volume kernels):
#   tools/mklib.sh NAME [-f file.hip] [-DX=Y ...]
# -> tools/ubench/libdfe_NAME.so (the other objects are the product build's); use with DFE_LIB=... / tools/ab2.sh / tools/ab_bench.sh
set -e
cd "$(dirname "$0")/../depth-estimation_amd/csrc"
name=$1; shift
src=ssd_cost_volume.hip
if [ "$1" = "-f" ]; then src=$2; shift 2; fi
make -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize "$@" -c $src -o /tmp/side_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/ubench/libdfe_$name.so /tmp/side_$name.o $objs -Wl,-soname,libdfe.so
echo tools/ubench/libdfe_$name.so
