O=gpurun_out/r3g; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for len in 1000 1400; do
RATTLE_HIP_LIB=$V/librattle_hip_barprof.so timeout 300 python tools/bench_poa_class.py $len 2560 200 0.10 1 2>&1 | grep -E "iter|barrier|wave 1" | sed "s/^/$len: /" >> $O/barprof.log
done
cat $O/barprof.log
