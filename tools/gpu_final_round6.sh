#!/bin/bash
# Final evidence of round 6 on the committed tree: full GPU suite, smoke, default bench (side configs, CPU baselines, toyset with its CPU
# ratio, no_stage), the same under rocprofv3 --kernel-trace --stats, the PMC passes of kernel C at 1e6 and 1e5 reads (per COMPUTED cell),
# the chain microbenchmarks, the CLI end to end.   usage: tools/gpu_final_round6.sh TAG
TAG=${1:-r6z}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $O/tests.log 2>&1; grep -n "passed\|failed\|Timeout" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python bench.py --steps 4 > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-configs > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
READS_PMC=1000000 bash tools/gpu_pmc_only.sh ${TAG}_pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-600
READS_PMC=100000 bash tools/gpu_pmc_only.sh ${TAG}_pmc100k > $O/pmc100k.log 2>&1; tail -2 $O/pmc100k.log | cut -c1-600
( python tools/bench_chain.py 1100 1 550 0.00002; python tools/bench_chain.py 1100 256 200 0.00002; python tools/bench_chain.py 1100 1792 200 0.00002; python tools/bench_chain.py 1100 4096 200 0.00002 ) 2>&1 | grep "^band" > $O/band_chain_microbench.txt
RATTLE_TIMING=1 timeout 900 bash tools/cli_e2e.sh 1000000 > $O/cli_e2e.txt 2>&1; grep -E "rattle c|wait for" $O/cli_e2e.txt
python - <<PY
import json
for f in ('bench_default','bench_under_rocprof'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']; print(f, round(d['value']), d.get('phases_ms_per_step'), r.get('gcups'), r.get('gcups_reference_cells'), r.get('frac'), r.get('frac_alg'), r.get('pmc_stale'), d.get('step_ms'))
        for k,v in (d.get('configs') or {}).items(): print('  ', k, {x: v.get(x) for x in ('value','ms_per_step','error')}, (v.get('roofline') or {}).get('gcups'), (v.get('roofline') or {}).get('frac'))
        if d.get('toyset'): print('   toyset', {x: d['toyset'].get(x) for x in ('cluster_s','correct_s','reads_per_s','clusters_equal_reference_fixture')}, (d['toyset'].get('cpu_all_cores') or {}).get('gpu_over_cpu'))
        print('   no_stage', d.get('no_stage'))
    except Exception as e: print(f, 'failed', e)
PY
