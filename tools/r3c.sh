O=gpurun_out/r3c; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for n in o8r6b; do
rm -f $O/tl_$n.txt
RATTLE_HIP_LIB=$V/librattle_hip_$n.so RATTLE_TIMING=1 RATTLE_POA_TIMELINE=$PWD/$O/tl_$n.txt timeout 600 python bench.py --no-cpu-baseline --warmup 1 > $O/bench_$n.json 2>$O/bench_$n.err
python tools/timeline_summary.py $O/tl_$n.txt 25 > $O/tl_$n.summary
cat $O/tl_$n.summary
grep -E "poa class|poa pass|stage" $O/bench_$n.err | tail -24
done
