#!/bin/bash
# Round-2 measurement pass on the GPU box: default bench, rocprofv3 kernel stats, PMC passes (separate runs, counters only with
# --kernel-trace), kernel C phase ticks (instrumented build), config 3 (--iso) and config 5 (mixed lengths) runs.
# usage: tools/gpu_profile_round2.sh TAG   (outputs under gpurun_out/TAG/)
set -x
TAG=${1:-r2p}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${READS_PMC:-300000}
# 1. default bench line (1e6 reads, with cpu baselines)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
# 2. kernel trace + stats of the default workload
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
# 3. PMC passes at $R reads (counters only; one group per run)
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  N=$(echo $G | cut -d' ' -f1)
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$N -- python $GRAFT_REPO_ROOT/bench.py --reads $R --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$N.json 2> $GRAFT_REPO_ROOT/$O/pmc_$N.err )
done
python tools/pmc_summary.py $O/pmc_poa.json $O/pmc_FETCH_SIZE.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU > $O/pmc_summary.log 2>&1
# 4. phase ticks of kernel C (instrumented library)
rm -f $O/poa_phase_ticks.jsonl
RATTLE_HIP_LIB=$PWD/rattle_amd/csrc/librattle_hip_prof.so RATTLE_POA_PROFILE_JSON=$PWD/$O/poa_phase_ticks.jsonl timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_prof_build.json 2> $O/bench_prof_build.err
# 5. config 3: 1e6 reads --iso
timeout 900 python bench.py --iso > $O/bench_iso.json 2> $O/bench_iso.err
# 6. config 5: mixed-length --rna reads
timeout 1500 python tools/run_mixed.py ${READS_MIXED:-500000} > $O/mixed.log 2>&1
ls -la $O
