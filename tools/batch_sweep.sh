#!/bin/bash
# One-off sweeps of the queue route's launch geometry on BASELINE config 4 (bench.py lines, ms per step and per kernel).
# Usage: tools/batch_sweep.sh batch | group
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items()})"; }
if [ "${1:-batch}" = batch ]; then
  for b in 16777216 33554432 67108864; do
    echo "share batch $b: $(DRT_HIP_BATCH_PATHS=$b python bench.py --config 4 --no-cpu-baseline --no-extra-views 2>/dev/null | line)"
  done
  for b in 16777216 67108864 134217728 268435456; do
    echo "full batch $b: $(DRT_HIP_BATCH_PATHS=$b python bench.py --config 4 --spp 256 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-views 2>/dev/null | line)"
  done
else
  for g in 1 2 4 8 16; do
    echo "share list group $g: $(DRT_HIP_SHADE_LIST_GROUP=$g python bench.py --config 4 --no-cpu-baseline --no-extra-views 2>/dev/null | line)"
  done
  for r in 128 256 512 1024; do
    echo "share region size $r: $(DRT_HIP_REGION_SIZE=$r python bench.py --config 4 --no-cpu-baseline --no-extra-views 2>/dev/null | line)"
  done
fi
