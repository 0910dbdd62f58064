#!/bin/bash
# the short form of tools/ab_edit_variants.sh: a lone pack on two teams and the device full of the 1536-column class in the barrier form,
# in-tree build against three "unrelated edit" builds, two repeats.  usage: ab_edit_variants_short.sh TAG
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for rep in 1 2; do
for n in base e1 e3 e4; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  r=$(RATTLE_HIP_LIB=$L RATTLE_POA_MODE=mt2 timeout 300 python tools/bench_poa_class.py 1000 1 200 0.10 2 2>/dev/null | tail -1); echo "$n mt2 packs 1: $r" | tee -a $O/ab_edit.log
  r=$(RATTLE_HIP_LIB=$L RATTLE_POA_MODE=dense timeout 300 python tools/bench_poa_class.py 1400 2048 200 0.10 2 2>/dev/null | tail -1); echo "$n dense1400 packs 2048: $r" | tee -a $O/ab_edit.log
done; done
