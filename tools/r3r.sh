O=gpurun_out/r3r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_edges.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(round(d['value']), d['phases_ms_per_step'], d['kernels_ms_per_step'], d['roofline']['gcups'])"
grep -E "host steps|job\(s\)|greedy driver|cluster_unsorted" $O/bench.err | tail -4
READS_PMC=300000 bash tools/gpu_pmc_only.sh r3r_pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
