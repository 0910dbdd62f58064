O=gpurun_out/s3i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
RATTLE_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep "correct: total\|cluster_unsorted\|build_index" $O/bench.err | tail -4
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['gcups'], d['checks'])"
