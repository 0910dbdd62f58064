O=gpurun_out/r3b; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for n in base o8r6 o7r8 o8r6b; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  for len in 1000 1400; do
    RATTLE_HIP_LIB=$L RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py $len 2560 2>&1 | grep -E "iter 1|poa class" | tail -2 | sed "s/^/$n $len: /" >> $O/micro.log
  done
  RATTLE_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['value']), round(d['roofline']['gcups'],1), d['phases_ms_per_step'], d['checks']['correct_digest'])" >> $O/bench.log
done
cat $O/micro.log $O/bench.log
