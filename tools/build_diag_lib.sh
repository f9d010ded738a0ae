#!/bin/bash
# A diagnostic build of the library beside the product one:  bash tools/build_diag_lib.sh <name> <file.hip> -DFLAG [...]
# recompiles ONE source with the extra flags, links it with the product objects of the others into tools/ab/libuavac_<name>.so
# (git-ignored; travels with gpurun).  Load it with UAVAC_LIB=tools/ab/libuavac_<name>.so.
set -eu
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
P=uav-autonomous-control_amd
make -C $P >/dev/null
mkdir -p tools/ab /tmp/uavac_diag
O=/tmp/uavac_diag/${NAME}_$(basename "$SRC" .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$P/csrc -Wall -Wno-unused-function "$@" -c "$P/csrc/$SRC" -o "$O"
OBJS=$(ls $P/build/*.o | grep -v "/$(basename "$SRC" .hip).o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/libuavac_${NAME}.so $OBJS "$O" -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libuavac.so
echo tools/ab/libuavac_${NAME}.so
