#!/bin/bash
# Tuning only: a side build of libdfe with extra macros for the cost-volume kernels:  tools/mklib.sh NAME [-DX=Y ...]
# -> tools/ubench/libdfe_NAME.so (the other objects are the product build's); use with DFE_LIB=... / tools/ab2.sh
set -e
cd "$(dirname "$0")/../depth-estimation_amd/csrc"
name=$1; shift
make -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize "$@" -c ssd_cost_volume.hip -o /tmp/ssd_cv_$name.o
objs=$(ls *.o | grep -v ssd_cost_volume.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/ubench/libdfe_$name.so /tmp/ssd_cv_$name.o $objs -Wl,-soname,libdfe.so
echo tools/ubench/libdfe_$name.so
