O=gpurun_out/r3e; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py tests/test_gpu_edges.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for n in base v3off v3o8; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  for len in 1000 1400; do
    RATTLE_HIP_LIB=$L RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py $len 2560 2>&1 | grep -E "iter 1|poa class" | tail -2 | sed "s/^/$n $len: /" >> $O/micro.log
  done
done
cat $O/micro.log
