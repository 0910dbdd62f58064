O=gpurun_out/r3k; mkdir -p $O
rm -f $O/tl.txt
RATTLE_TIMING=1 RATTLE_POA_TIMELINE=$PWD/$O/tl.txt timeout 900 python tools/run_mixed.py 100000 > $O/mixed.log 2> $O/mixed.err
python tools/timeline_summary.py $O/tl.txt 20 > $O/tl.summary
tail -2 $O/mixed.log | cut -c1-600; grep -E "poa class|poa pass|correct: stage|cluster" $O/mixed.err | head -60; cat $O/tl.summary | head -60
