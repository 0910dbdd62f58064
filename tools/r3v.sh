O=gpurun_out/r3v; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_cli.py -x -q -k "longer or mixed_length_flow or absurd" > $O/tests.log 2>&1; tail -4 $O/tests.log
rm -f $O/tl.txt
RATTLE_TIMING=1 RATTLE_POA_TIMELINE=$PWD/$O/tl.txt timeout 1500 python tools/run_mixed.py 100000 20000 > $O/mixed.log 2> $O/mixed.err
python tools/timeline_summary.py $O/tl.txt 20 > $O/tl.summary
tail -1 $O/mixed.log | cut -c1-420; head -11 $O/tl.summary
