#!/bin/bash
# Final evidence of round 4 on the committed tree: full GPU suite, smoke, default bench (with the side configs and CPU baselines),
# the same under rocprofv3 --kernel-trace --stats, the three PMC passes of kernel C AT THE BENCH'S SIZE (counters only, one group
# per pass), config 3 with its two PMC passes, kernel A's SQ counters.   usage: tools/gpu_final_round4.sh TAG
TAG=${1:-r4f}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
READS_PMC=${READS_PMC:-1000000} bash tools/gpu_pmc_only.sh ${TAG}_pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-400
RATTLE_TIMING=1 timeout 900 python bench.py --iso --no-cpu-baseline > $O/bench_iso.json 2> $O/bench_iso.err
READS_PMC=${READS_PMC:-1000000} bash tools/gpu_pmc_iso.sh ${TAG}_pmc_iso > $O/pmc_iso.log 2>&1; tail -2 $O/pmc_iso.log | cut -c1-300
bash tools/gpu_pmc_cluster.sh ${TAG}_pmc_cluster > $O/pmc_cluster.log 2>&1; tail -4 $O/pmc_cluster.log | cut -c1-600
python -c "
import json
for f in ('bench_default','bench_under_rocprof','bench_iso'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value']), d.get('phases_ms_per_step'), d['kernels_ms_per_step'], d['roofline'].get('gcups'), d['roofline'].get('frac'), d['roofline'].get('pmc_stale'))
        for k,v in (d.get('configs') or {}).items(): print('  ', k, {x: v.get(x) for x in ('value','ms_per_step','error')}, (v.get('roofline') or {}).get('gcups'), (v.get('roofline') or {}).get('frac'))
        if d.get('toyset'): print('   toyset', {x: d['toyset'].get(x) for x in ('cluster_s','correct_s','reads_per_s','clusters_equal_reference_fixture')})
    except Exception as e: print(f, 'failed', e)
"
RATTLE_TIMING=1 timeout 900 bash tools/cli_e2e.sh 1000000 > $O/cli_e2e.txt 2>&1; tail -32 $O/cli_e2e.txt
