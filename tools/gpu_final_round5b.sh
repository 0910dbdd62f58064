#!/bin/bash
# Second half of round 5's final evidence (after the head-start criterion was narrowed to a handful of workgroups): GPU suite, three default
# benches with stage times, rank replay at 8, PMC passes at 1e6 and 1e5 reads (the counters' source hash must match the tree), rocprof stats.
TAG=${1:-r5final2}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 > $O/tests.log 2>&1; grep -n "passed\|failed\|Timeout" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python tools/rank_replay.py --reads 1000000 --worlds 2,4,8 --out "$O/rank_replay_{}.json" > $O/rank_replay.log 2>&1; tail -1 $O/rank_replay.log | cut -c1-400
READS_PMC=1000000 bash tools/gpu_pmc_only.sh ${TAG}_pmc > $O/pmc.log 2>&1; tail -1 $O/pmc.log | cut -c1-300
READS_PMC=100000 bash tools/gpu_pmc_only.sh ${TAG}_pmc100k > $O/pmc100k.log 2>&1; tail -1 $O/pmc100k.log | cut -c1-300
cp gpurun_out/${TAG}_pmc/pmc_poa.json profiles/round5_pmc_poa.json; cp gpurun_out/${TAG}_pmc100k/pmc_poa.json profiles/round5_pmc_poa_100k.json
RATTLE_TIMING=1 timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
for i in 1 2; do RATTLE_TIMING=1 python bench.py --no-cpu-baseline > $O/bench_repeat$i.json 2> $O/bench_repeat$i.err; done
python -c "
import json
for f in ('bench_default','bench_under_rocprof','bench_repeat1','bench_repeat2'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']; print(f, round(d['value']), d.get('phases_ms_per_step'), r.get('gcups'), r.get('frac'), r.get('frac_practical'), r.get('salu_frac'), r.get('pmc_stale'), d.get('step_ms'))
        for k,v in (d.get('configs') or {}).items(): print('  ', k, {x: v.get(x) for x in ('value','ms_per_step','error')}, (v.get('roofline') or {}).get('gcups'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('pmc_stale'))
        if d.get('toyset'): print('   toyset', {x: d['toyset'].get(x) for x in ('cluster_s','correct_s','reads_per_s','clusters_equal_reference_fixture')})
    except Exception as e: print(f, 'failed', e)
"
grep -h "correct: stage" $O/bench_default.err $O/bench_repeat1.err $O/bench_repeat2.err | awk '{printf "%s ", $(NF-1)} END {print ""}'
