O=gpurun_out/r3w; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for n in base lr0 lr3p lr0p; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  RATTLE_HIP_LIB=$L RATTLE_TIMING=1 timeout 600 python tools/bench_poa_class.py 12000 8 24 0.10 1 2>&1 | grep -E "iter|phases|profile" | sed "s/^/$n: /" | cut -c1-330 >> $O/long.log
done
cat $O/long.log
