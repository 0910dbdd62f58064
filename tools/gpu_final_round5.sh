#!/bin/bash
# Final evidence of round 5 on the committed tree: full GPU suite (every test bounded), smoke, default bench (with the side configs and
# CPU baselines), the same under rocprofv3 --kernel-trace --stats, the PMC passes of kernel C at the bench's size AND at 1e5 reads
# (the under-filled device runs other forms of the row loop), config 3 with its PMC passes, the CLI end to end.
# usage: tools/gpu_final_round5.sh TAG
TAG=${1:-r5z}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 > $O/tests.log 2>&1; grep -n "passed\|failed\|Timeout" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
READS_PMC=1000000 bash tools/gpu_pmc_only.sh ${TAG}_pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-600
READS_PMC=100000 bash tools/gpu_pmc_only.sh ${TAG}_pmc100k > $O/pmc100k.log 2>&1; tail -2 $O/pmc100k.log | cut -c1-600
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats100k -- python $GRAFT_REPO_ROOT/bench.py --reads 100000 --steps 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof_100k.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof_100k.err )
RATTLE_TIMING=1 timeout 900 python bench.py --iso --no-cpu-baseline > $O/bench_iso.json 2> $O/bench_iso.err
READS_PMC=1000000 bash tools/gpu_pmc_iso.sh ${TAG}_pmc_iso > $O/pmc_iso.log 2>&1; tail -2 $O/pmc_iso.log | cut -c1-300
python -c "
import json
for f in ('bench_default','bench_under_rocprof','bench_under_rocprof_100k','bench_iso'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']; print(f, round(d['value']), d.get('phases_ms_per_step'), d['kernels_ms_per_step'], r.get('gcups'), r.get('frac'), r.get('frac_practical'), r.get('salu_frac'), r.get('pmc_stale'), d.get('step_ms'))
        for k,v in (d.get('configs') or {}).items(): print('  ', k, {x: v.get(x) for x in ('value','ms_per_step','error')}, (v.get('roofline') or {}).get('gcups'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('pmc_source'))
        if d.get('toyset'): print('   toyset', {x: d['toyset'].get(x) for x in ('cluster_s','correct_s','reads_per_s','clusters_equal_reference_fixture')})
    except Exception as e: print(f, 'failed', e)
"
RATTLE_TIMING=1 timeout 900 bash tools/cli_e2e.sh 1000000 > $O/cli_e2e.txt 2>&1; grep -E "rattle c|wait for" $O/cli_e2e.txt
