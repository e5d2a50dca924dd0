#!/bin/bash
set -u
O=gpurun_out/${1:-st}; mkdir -p $O
timeout 1200 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
tail -5 $O/tests.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-views --absorb 0.5 --min-bounces 1 > $O/bench_rr.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-views --config 5 > $O/bench_c5.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-views --config 4 > $O/bench_c4.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-views --scene mesh160x160 --unbiased > $O/bench_unb_mesh.json 2>> $O/bench.err
python3 - <<PY
import json
for f in ["bench","bench_rr","bench_c5","bench_c4","bench_unb_mesh"]:
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("program"), d.get("serial_frame"), (d.get("generic_program") or {}).get("ms_per_step"), (d.get("f64") or {}).get("ms_per_step"), (d.get("unbiased") or {}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/bench.err
