#!/bin/bash
# Where do the shade / backward kernels' cycles go: per-pipe active cycles, instruction mix, lane use,
# scalar-memory latency.  Run through gpurun.  Usage: tools/pmc_issue.sh [bench args]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_issue
mkdir -p "$OUT"
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU" \
           "SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVES"; do
  i=$((i+1))
  d=$OUT/set$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$d" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-views "$@" > /dev/null 2> "$d.err" || { echo "bench.py failed under rocprofv3 (set $i): see $d.err" >&2; tail -5 "$d.err" >&2; exit 1; }
  python3 - "$d" <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no csv in", sys.argv[1]); sys.exit()
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:28]
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in d:
    if any(x in k for x in os.environ.get("PMC_KERNELS", "k_shade,k_backward").split(",")):
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d[k].items()})
PY
done
