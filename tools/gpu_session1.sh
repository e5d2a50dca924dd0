#!/bin/bash
# first GPU session of round 4: baseline at HEAD, integer-multiply issue cost, hiprtc on the box
set -u
O=gpurun_out/s1; mkdir -p $O
build/microbench_imul > $O/imul.txt 2>&1
build/hiprtc_probe tools/jit_probe_src.hip "" -Ibuild/jit_hdrs > $O/hiprtc.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench_head.json 2> $O/bench_head.err
DRT_HIP_OVERLAP_FRAMES=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-views > $O/bench_serial.json 2>&1
DRT_HIP_JIT=-1 DRT_HIP_OVERLAP_FRAMES=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-views > $O/bench_generic.json 2>&1
cat $O/imul.txt $O/hiprtc.txt
