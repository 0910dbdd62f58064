O=gpurun_out/r3i; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for n in e6m5 e4m6 e5m5; do
for len in 1000 1400; do
RATTLE_POA_EXP=1,2 RATTLE_HIP_LIB=$V/librattle_hip_$n.so RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py $len 2560 200 0.10 2 2>&1 | grep -E "iter 1|poa class" | tail -2 | sed "s/^/$n $len: /" >> $O/micro.log
done; done
cat $O/micro.log
