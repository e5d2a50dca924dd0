#!/bin/bash
set -u
O=gpurun_out/${1:-sab}; mkdir -p $O
AB_ROUNDS=21 python3 tools/ab_libs.py build/lib_5w.so build/lib_6w.so differentiable-renderer_amd/libdrt_hip.so > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 900 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
tail -3 $O/tests.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-views --absorb 0.5 --min-bounces 1 > $O/bench_rr.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-views --config 5 > $O/bench_c5.json 2>> $O/bench.err
python3 - <<PY
import json
for f in ["bench","bench_rr","bench_c5"]:
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("serial_frame"), d.get("generic_program"), d.get("f64"), d.get("fwd_only"), d.get("unbiased"))
PY
