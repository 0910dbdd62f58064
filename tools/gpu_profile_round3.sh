#!/bin/bash
# Round-3 evidence pass on the GPU box: full GPU test suite, smoke, default bench, rocprofv3 kernel stats, PMC passes (separate
# runs, counters only with --kernel-trace), config 3 (--iso), config 5 (mixed lengths).  usage: tools/gpu_profile_round3.sh TAG [MIXED_READS]
TAG=${1:-r3}; MIXED=${2:-500000}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
READS_PMC=300000 bash tools/gpu_pmc_only.sh ${TAG}_pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-400
timeout 900 python bench.py --iso --no-cpu-baseline > $O/bench_iso.json 2> $O/bench_iso.err
timeout 1500 python tools/run_mixed.py $MIXED 20000 > $O/mixed.log 2> $O/mixed.err; tail -1 $O/mixed.log | cut -c1-900
RATTLE_TIMING=1 timeout 900 bash tools/cli_e2e.sh 1000000 > $O/cli_e2e.txt 2>&1; tail -25 $O/cli_e2e.txt
python -c "
import json
for f in ('bench_default','bench_under_rocprof','bench_iso'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value']), d.get('phases_ms_per_step'), d['roofline'].get('gcups'), d['roofline'].get('frac'), d['roofline'].get('pmc_stale'))
    except Exception as e: print(f, 'failed', e)
"
