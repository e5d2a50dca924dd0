#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile.sh <tag> <workload key> [bench args...]     outputs under gpurun_out/prof_<tag>/
#   workload key = what bench.py compares profiles/traffic.json against, "<scene>:<W>x<H>x<spp>:d<depth>:<fwd|fwdbwd>"
#   (the headline: cornell:512x512x64:d8:fwdbwd)
set -u
TAG=${1:-r02}; shift || true
WORKLOAD=${1:-cornell:512x512x64:d8:fwdbwd}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# Every pass with the frames in stream order: with overlapping frames (the default of device-pointer renders that do not wait)
# the k_path grids of two consecutive frames run side by side and a launch's begin-to-end span in the kernel trace is no longer
# its own duration (round 3's trace: avg 1387 us for a 790 us kernel).  bench.py's roofline describes one launch at a time,
# measured live with HIP events; this trace must agree with it.  (Exported here, not through `env` on the profiler's
# command line: the profiler's preloaded library has initialised the GPU by then and an exec hop is refused.)
export DRT_HIP_OVERLAP_FRAMES=0
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-extra-views $*"
cd "$PWD"
# (every profiler run under `timeout`: a run that aborts has been seen to hang until the box is taken away)
# 1) per-kernel time
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
# 2) HBM traffic counters, one pass each (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_write.err"
# 3) wave-level counters
timeout -k 10 900 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_sq.err"
python3 tools/summarize_profile.py "$OUT" "$WORKLOAD" > "$OUT/summary.txt" 2> "$OUT/summary.err"
cat "$OUT/summary.txt"
# keep only small files in gpurun_out (it is merged back, <= 64 MiB)
find "$OUT" -name "*.csv" -size +8M -delete
