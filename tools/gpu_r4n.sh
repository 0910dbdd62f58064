#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
for v in base psold base; do
  L=$PWD/rattle_amd/csrc/librattle_hip.so; [ $v = psold ] && L=$PWD/rattle_amd/csrc/variants/librattle_hip_psold.so
  RATTLE_HIP_LIB=$L RATTLE_TIMING=1 timeout 600 python tools/run_mixed.py 200000 20000 > $O/mixed_$v.log 2> $O/mixed_$v.err
  echo "== $v: $(tail -1 $O/mixed_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cluster_s'], d['correct_s'], d['polish_s'])")"
  grep -E "job\(s\)|gather \+ build|filter|count pass|full pass" $O/mixed_$v.err | head -4 | cut -c1-300
done
