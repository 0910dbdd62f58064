#!/bin/bash
TAG=${1:-r5c}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_poa.py -x -q -m gpu --timeout 90 > $O/tests_poa.log 2>&1; echo "poa tests: $(tail -1 $O/tests_poa.log)"; grep -n "Error\|error\|FAILED\|Timeout" $O/tests_poa.log | head -20
