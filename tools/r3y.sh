O=gpurun_out/r3y; mkdir -p $O
RATTLE_TIMING=1 timeout 900 python bench.py --iso --no-cpu-baseline > $O/iso.json 2> $O/iso.err
grep -E "job\(s\)|host steps|greedy|iso|cluster" $O/iso.err | tail -14
python -c "
import json; d=json.loads(open('$O/iso.json').read().strip().splitlines()[-1]); print(d['value'], d['kernels_ms_per_step'], d['cluster_counters'], d['roofline'])"
