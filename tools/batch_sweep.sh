for b in 16777216 33554432 67108864; do
  echo "share batch $b: $(DRT_HIP_BATCH_PATHS=$b python bench.py --config 4 --no-cpu-baseline --no-extra-views 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items()})")"
done
for b in 16777216 67108864 134217728 268435456; do
  echo "full batch $b: $(DRT_HIP_BATCH_PATHS=$b python bench.py --config 4 --spp 256 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-views 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items()})")"
done
