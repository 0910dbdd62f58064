O=gpurun_out/r3u; mkdir -p $O
rm -f $O/tl.txt
RATTLE_TIMING=1 RATTLE_POA_TIMELINE=$PWD/$O/tl.txt timeout 1500 python tools/run_mixed.py 500000 20000 > $O/mixed.log 2> $O/mixed.err
python tools/timeline_summary.py $O/tl.txt 20 > $O/tl.summary
grep -E "poa class|poa pass|correct: stage|correct: total|cluster_unsorted|greedy driver|job\(s\)|host steps|build_index" $O/mixed.err | head -50; head -24 $O/tl.summary
