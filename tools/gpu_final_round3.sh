#!/bin/bash
# Final evidence of round 3 on the committed tree: full GPU suite, smoke, default bench (with the CPU baselines), the same under
# rocprofv3 --kernel-trace --stats, the three PMC passes (counters only), config 3.  usage: tools/gpu_final_round3.sh TAG
TAG=${1:-r3e}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
READS_PMC=300000 bash tools/gpu_pmc_only.sh ${TAG}_pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-400
timeout 900 python bench.py --iso --no-cpu-baseline > $O/bench_iso.json 2> $O/bench_iso.err
python -c "
import json
for f in ('bench_default','bench_under_rocprof','bench_iso'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value']), d.get('phases_ms_per_step'), d['kernels_ms_per_step'], d['roofline'].get('gcups'), d['roofline'].get('frac'), d['roofline'].get('pmc_stale'))
    except Exception as e: print(f, 'failed', e)
"
