O=gpurun_out/r3x; mkdir -p $O
echo skip tests
for n in 500000; do
rm -f $O/tl_$n.txt
RATTLE_TIMING=1 RATTLE_POA_TIMELINE=$PWD/$O/tl_$n.txt timeout 1500 python tools/run_mixed.py $n 20000 > $O/mixed_$n.log 2> $O/mixed_$n.err
python tools/timeline_summary.py $O/tl_$n.txt 20 > $O/tl_$n.summary
tail -1 $O/mixed_$n.log | cut -c1-760; head -11 $O/tl_$n.summary; grep -E "poa class|poa pass|correct: stage" $O/mixed_$n.err | head -14
done
