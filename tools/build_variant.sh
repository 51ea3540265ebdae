#!/bin/bash
# A/B builds of the step kernels: tools/build_variant.sh <name> [-DMACRO=V ...]
#   -> graph-hscn_amd/graph_hscn/lib/libhscn_<name>.so = the shipped objects with resident.o rebuilt under the flags
set -e
NAME=$1; shift
UNIT=${UNIT:-resident}          # which translation unit the flags rebuild (UNIT=resident_scn tools/build_variant.sh ...)
cd $(dirname $0)/../graph-hscn_amd
mkdir -p build/var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I../include "$@" -c csrc/$UNIT.hip -o build/var_$NAME/$UNIT.o
OBJS=$(ls build/*.o | grep -v "build/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o graph_hscn/lib/libhscn_$NAME.so $OBJS build/var_$NAME/$UNIT.o
echo built libhscn_$NAME.so
