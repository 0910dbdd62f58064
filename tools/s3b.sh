O=gpurun_out/s3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_edges.py tests/test_gpu_dist.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
RATTLE_TIMING=1 timeout 900 python bench.py --iso --no-cpu-baseline > $O/iso.json 2> $O/iso.err
grep -E "job\(s\)|host steps|greedy|iso|cluster" $O/iso.err | tail -14
python -c "
import json; d=json.loads(open('$O/iso.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['cluster_counters'])"
RATTLE_TIMING=1 timeout 600 python tools/cluster_only.py > $O/cluster_only.log 2>&1; tail -25 $O/cluster_only.log
