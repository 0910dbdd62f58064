O=gpurun_out/r3d; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
rocm-smi --showclocks --showpower > $O/smi_idle.txt 2>&1
for n in base o8r6b; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  ( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi_$n.txt &
  SMI=$!
  RATTLE_HIP_LIB=$L RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --warmup 1 > $O/bench_$n.json 2>$O/bench_$n.err
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  grep -E "stage 1|stage 2a|stage 2b|correct: total" $O/bench_$n.err | tail -4
done
head -5 $O/smi_idle.txt; sed -n '20,50p' $O/smi_base.txt | cut -c1-200
