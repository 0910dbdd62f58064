#!/bin/bash
# Team kernels (two and four teams) and the barrier form on the in-tree build and on the four "unrelated edit" builds of
# tools/build_edit_variants.sh, interleaved on ONE box: a lone 200-read pack and one pack per CU.  usage: ab_edit_variants.sh TAG [REPS]
TAG=$1; REPS=${2:-3}
O=gpurun_out/$TAG; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for rep in $(seq $REPS); do
for n in base e1 e2 e3 e4; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  for mode in mt2 mt4 dense; do
    for packs in 1 256; do
      r=$(RATTLE_HIP_LIB=$L RATTLE_POA_MODE=$mode timeout 300 python tools/bench_poa_class.py 1000 $packs 200 0.10 2 2>/dev/null | tail -1)
      echo "$n $mode packs $packs: $r" | tee -a $O/ab_edit.log
    done
  done
  # the device full of the 1024- and 1536-column classes in the barrier form (stage 1 of `correct`): the 1536 class is held to 64 vector
  # registers with ~24 of them spilled -- where the reloads land is the allocator's choice
  for len in 1000 1400; do
    r=$(RATTLE_HIP_LIB=$L RATTLE_POA_MODE=dense timeout 300 python tools/bench_poa_class.py $len 2048 200 0.10 2 2>/dev/null | tail -1)
    echo "$n dense$len packs 2048: $r" | tee -a $O/ab_edit.log
  done
done; done
python - $O/ab_edit.log <<'PY'
import re, sys, collections
t = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"(\w+) (\w+) packs (\d+): .*kernel (\d+) ms", l)
    if m: t[(m.group(2), int(m.group(3)), m.group(1))].append(int(m.group(4)))
for mode in ("mt2", "mt4", "dense", "dense1000", "dense1400"):
    for packs in (1, 256, 2048):
        best = {n: min(t[(mode, packs, n)]) for n in ("base", "e1", "e2", "e3", "e4") if t[(mode, packs, n)]}
        if best:
            lo, hi = min(best.values()), max(best.values())
            print(f"{mode} packs {packs}: best of the repeats per build {best}  spread {100.0 * (hi - lo) / lo:.1f} %")
PY
