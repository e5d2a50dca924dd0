mkdir -p gpurun_out/r05
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests_5.txt 2>&1
tail -8 gpurun_out/r05/gputests_5.txt
timeout 600 python bench.py --single-process --gpus 2 --same-gpu --steps 5 --warmup 1 2>&1 | tail -1 | cut -c1-900
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-extra-views --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['config']['parallelism'], d['config']['gpu_max_hw_queues']); print(d['roofline'].get('pmc_note'), d['roofline']['frac'])"
