set -x
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests/test_gpu_dist.py -x -q -k "bench" > $O/test_bench.log 2>&1
RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
for L in 1000 1400; do
timeout 300 python tools/bench_poa_class.py $L 2560 > $O/poa_${L}_default.log 2>&1
RATTLE_POA_EXP=0,0 timeout 300 python tools/bench_poa_class.py $L 2560 > $O/poa_${L}_exp0.log 2>&1
RATTLE_POA_EXP=1,2 timeout 300 python tools/bench_poa_class.py $L 2560 > $O/poa_${L}_exp12.log 2>&1
done
