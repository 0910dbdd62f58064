O=gpurun_out/r3h; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for e in "" "1,2" "0,0"; do
for len in 1000 1400; do
RATTLE_POA_EXP=$e RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py $len 2560 200 0.10 2 2>&1 | grep -E "iter 1|poa class" | tail -2 | sed "s/^/exp[$e] $len: /" >> $O/micro.log
RATTLE_POA_EXP=$e RATTLE_HIP_LIB=$V/librattle_hip_prof.so timeout 300 python tools/bench_poa_class.py $len 2560 200 0.10 1 2>&1 | grep -E "phases|profile" | sed "s/^/exp[$e] $len: /" >> $O/micro.log
done; done
cat $O/micro.log
