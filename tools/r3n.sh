O=gpurun_out/r3n; mkdir -p $O
for q in 8; do
rm -f $O/tl_$q.txt
GPU_MAX_HW_QUEUES=$q RATTLE_POA_STREAMS=8 RATTLE_POA_TIMELINE=$PWD/$O/tl_$q.txt timeout 1500 python tools/run_mixed.py 100000 20000 > $O/mixed_$q.log 2> $O/mixed_$q.err
python tools/timeline_summary.py $O/tl_$q.txt 20 > $O/tl_$q.summary
tail -1 $O/mixed_$q.log | cut -c1-420; head -11 $O/tl_$q.summary
done
