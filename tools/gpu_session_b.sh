#!/bin/bash
set -u
O=gpurun_out/${1:-sb}; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 10 --warmup 3 --scene random11 --no-cpu-baseline --no-extra-views > $O/bench_random11.json 2> $O/bench_random11.err
python3 bench.py --steps 5 --warmup 2 --config 4 --no-cpu-baseline --no-extra-views > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
(python3 tools/fit_albedo.py --quiet; python3 tools/fit_albedo.py --quiet --async) > $O/fit.txt 2>&1
tail -3 $O/tests.txt; cat $O/fit.txt; tail -2 $O/bench.err
python3 - <<PY
import json
for f in ["bench","bench_random11","bench_c4"]:
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("mode"), d["config"].get("specialise_ms"), d.get("serial_frame"), d.get("generic_program"), d["roofline"].get("frac"), d["roofline"].get("avg_launch_ms"), d["roofline"].get("pipelined"))
    except Exception as e: print(f, "ERR", e)
PY
