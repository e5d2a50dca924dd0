#!/bin/bash
set -u
O=gpurun_out/${1:-ssc}; mkdir -p $O
AB_ROUNDS=21 python3 tools/ab_libs.py build/lib_poly.so differentiable-renderer_amd/libdrt_hip.so > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 1200 python3 -m pytest tests -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
tail -6 $O/tests.txt
python3 tools/parity_report.py --big > $O/parity.txt 2>&1; tail -12 $O/parity.txt
python3 tools/fuzz_reference.py 150 11 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; cat $O/smoke.txt
