#!/bin/bash
# build a variant of librattle_hip.so with extra -D flags for poa.hip: tools/build_variant.sh NAME -DPOA_MW_4x4=8 ...
# -> rattle_amd/csrc/variants/librattle_hip_NAME.so (select with RATTLE_HIP_LIB)
set -e
cd "$(dirname "$0")/../rattle_amd/csrc"
NAME=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result "$@" -c poa.hip -o variants/poa_$NAME.o
OBJS=$(ls *.o | grep -v '^poa' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/librattle_hip_$NAME.so $OBJS variants/poa_$NAME.o -lpthread -ldl
echo built variants/librattle_hip_$NAME.so
