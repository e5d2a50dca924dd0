#!/bin/bash
set -u
O=gpurun_out/${1:-sj}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_jit.py -x -q > $O/tests_jit.txt 2>&1; echo "rc=$?" >> $O/tests_jit.txt
tail -15 $O/tests_jit.txt
python3 tools/ab_env.py --scene random11 "DRT_HIP_JIT=0" "DRT_HIP_JIT=force" "DRT_HIP_JIT=0" "DRT_HIP_JIT=force" > $O/ab_random11.txt 2>&1
python3 tools/ab_env.py --scene cornell "DRT_HIP_JIT=-1" "DRT_HIP_JIT=0" > $O/ab_cornell.txt 2>&1
cat $O/ab_random11.txt $O/ab_cornell.txt
timeout 900 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
tail -5 $O/tests.txt
