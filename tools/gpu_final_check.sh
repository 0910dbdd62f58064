#!/bin/bash
# Last check of a round on the committed tree: full GPU suite, smoke, default bench (CPU baselines included), config 3 (--iso),
# config 2 (1e5 reads), the sharded form on one device (two ranks, host transport).  usage: tools/gpu_final_check.sh TAG
TAG=${1:-final}
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --iso --no-cpu-baseline > $O/bench_iso.json 2> $O/bench_iso.err
timeout 600 python bench.py --reads 100000 --no-cpu-baseline > $O/bench_100k.json 2> $O/bench_100k.err
python -c "
import json
for f in ('bench_default','bench_iso','bench_100k'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],1), d.get('phases_ms_per_step'), d['kernels_ms_per_step'], d['roofline'].get('gcups'), d['roofline'].get('frac'), d['roofline'].get('pmc_stale'))
    except Exception as e: print(f, 'failed', e)
"
