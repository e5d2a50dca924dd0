#!/bin/bash
# generic GPU session: tests, then the three bench views (pipelined / serial / generic program)
set -u
O=gpurun_out/${1:-s}; mkdir -p $O
build/microbench_imul > $O/imul.txt 2>&1
timeout 900 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-views > $O/bench.json 2> $O/bench.err
DRT_HIP_OVERLAP_FRAMES=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-views > $O/bench_serial.json 2>&1
DRT_HIP_JIT=-1 DRT_HIP_OVERLAP_FRAMES=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-views > $O/bench_generic.json 2>&1
tail -5 $O/tests.txt; cat $O/imul.txt
for f in bench bench_serial bench_generic; do python3 -c "
import sys,json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['cpu_baseline'])"; done
