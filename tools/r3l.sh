O=gpurun_out/r3l; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
run() { # name lib exp
  rm -f $O/tl_$1.txt
  RATTLE_POA_EXP=$3 RATTLE_HIP_LIB=$2 RATTLE_POA_TIMELINE=$PWD/$O/tl_$1.txt timeout 600 python bench.py --no-cpu-baseline --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['roofline']['gcups'],1), d['phases_ms_per_step'], d['checks']['correct_digest'])" >> $O/bench.log
  python tools/timeline_summary.py $O/tl_$1.txt 10 | grep -A2 "^pass 4" | head -3 | cut -c1-200 >> $O/bench.log
}
B=$PWD/rattle_amd/csrc/librattle_hip.so
run base $B ""
run v3off $V/librattle_hip_v3off.so ""
run o7 $V/librattle_hip_o7.so ""
run v3o8 $V/librattle_hip_v3o8.so ""
run e4m6 $V/librattle_hip_e4m6.so "1,2"
run b2x8r8 $B "1,2"
run b1x16 $B "0,-1"
run base2 $B ""
cat $O/bench.log
