#!/bin/bash
# Four builds of kernel C with UNRELATED edits (copies of poa.hip, never the tracked file): does the speed of the team kernels depend on
# where an edit elsewhere pushes the register allocator?  ("a lone pack on two teams within +-3 % over five builds").
#   e1: a counter bumped in add_alignment under a debug bit nobody sets      e2: an extra (unused) device helper in front of the scans
#   e3: both, plus a second debug-only branch in the traceback's prologue    e4: the tie search's barrier comment turned into a real (dead) branch
# -> rattle_amd/csrc/variants/librattle_hip_e{1..4}.so; time them beside the in-tree build with tools/ab_edit_variants.sh on the GPU box
set -e
cd "$(dirname "$0")/../rattle_amd/csrc"
mkdir -p variants
E1='s|^            const uint32_t n_old = S.n_nodes;|            if (A.debug \& (1u << 21)) atomicAdd(\&A.counters[15], 1ull);\n            const uint32_t n_old = S.n_nodes;|'
E2='s|^__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {|__device__ __forceinline__ uint32_t unrelated_mix(uint32_t v) { return (v * 2654435761u) ^ (v >> 13); }\n__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {|'
E3='s|^                    if (tid == 0) s_bc\[5\] = 0xFFFFFFFFu;|                    if ((A.debug \& (1u << 22)) \&\& tid == 1) atomicAdd(\&A.counters[15], (unsigned long long)best);\n                    if (tid == 0) s_bc[5] = 0xFFFFFFFFu;|'
E4='s|^            enum { K_SKIP = 0, K_INS = 1, K_SAME = 2, K_SIB = 4, K_NEW = 3 };|            if ((A.debug \& (1u << 23)) \&\& n_aln > 100000u) { S.err = POA_ERR_ALN; }\n            enum { K_SKIP = 0, K_INS = 1, K_SAME = 2, K_SIB = 4, K_NEW = 3 };|'
build() {
  n=$1; shift
  cp poa.hip variants/poa_$n.hip
  for e in "$@"; do sed -i "$e" variants/poa_$n.hip; done
  cmp -s poa.hip variants/poa_$n.hip && { echo "variant $n: the edit did not apply"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I. -I../../include -c variants/poa_$n.hip -o variants/poa_$n.o 2>/dev/null
  OBJS=$(ls *.o | grep -v '^poa' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/librattle_hip_$n.so $OBJS variants/poa_$n.o -lpthread -ldl 2>/dev/null
  echo built variants/librattle_hip_$n.so
}
build e1 "$E1" &
build e2 "$E2" &
wait
build e3 "$E1" "$E2" "$E3" &
build e4 "$E4" &
wait
