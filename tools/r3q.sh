O=gpurun_out/r3q; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for rep in 1 2; do
for n in base c1a c1b c1c c1d c1e; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  for len in 1400 1000; do
    RATTLE_HIP_LIB=$L RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py $len 2560 2>&1 | grep -E "iter 1|poa class" | tail -2 | tr '\n' ' ' | sed "s/^/$n $len: /" >> $O/micro.log; echo >> $O/micro.log
  done
done; done
cut -c1-230 $O/micro.log
