O=gpurun_out/r3m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_poa.py tests/test_gpu_edges.py tests/test_gpu_correct.py tests/test_gpu_scale_properties.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
rm -f $O/tl.txt
RATTLE_TIMING=1 RATTLE_POA_TIMELINE=$PWD/$O/tl.txt timeout 1500 python tools/run_mixed.py 100000 20000 > $O/mixed.log 2> $O/mixed.err
python tools/timeline_summary.py $O/tl.txt 20 > $O/tl.summary
tail -1 $O/mixed.log | cut -c1-700; grep -E "poa class|poa pass|correct: stage" $O/mixed.err | head -24; head -12 $O/tl.summary
