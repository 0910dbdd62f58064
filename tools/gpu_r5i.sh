#!/bin/bash
TAG=${1:-r5i}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py -x -q -m gpu --timeout 150 > $O/tests.log 2>&1; echo "poa + correct tests: $(tail -1 $O/tests.log)"; grep -n "Error\|FAILED\|Timeout" $O/tests.log | head
for mode in dense mt1 mt4; do
  echo "== lone pack phases, $mode:"; RATTLE_HIP_LIB=$PWD/rattle_amd/csrc/librattle_hip_prof.so RATTLE_POA_MODE=$mode timeout 200 python tools/bench_poa_class.py 980 1 200 0.10 2 2>&1 | grep -E "phases|profile|^iter" | tail -3
done 2>&1 | tee $O/lone_phases.log
echo "== dense, device full:"; RATTLE_HIP_LIB=$PWD/rattle_amd/csrc/librattle_hip_prof.so RATTLE_POA_MODE=dense timeout 300 python tools/bench_poa_class.py 980 2560 200 0.10 2 2>&1 | grep -E "phases|profile|^iter" | tail -3 | tee $O/dense_phases.log
for l in 980 1450; do echo "== $l x 2560 dense: $(timeout 300 python tools/bench_poa_class.py $l 2560 200 0.10 2 2>&1 | grep -E "^iter" | tail -1)"; done | tee $O/dense_full.log
timeout 2300 python tools/rank_replay.py --reads 1000000 --worlds 2,4,8 --out "$O/round5_rank_replay_{}.json" > $O/replay_log.txt 2>&1; tail -1 $O/replay_log.txt | cut -c1-200
