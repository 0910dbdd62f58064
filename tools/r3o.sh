O=gpurun_out/r3o; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_correct.py tests/test_gpu_dist.py -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
