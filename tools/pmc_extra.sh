#!/bin/bash
# Extra PMC passes (LDS, L2, TCP) for K2/K3/K6; run through gpurun. Usage: tools/pmc_extra.sh [bench args]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_extra
mkdir -p "$OUT"
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
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
    if any(x in k for x in os.environ.get("PMC_KERNELS", "k_shade,k_intersect,k_backward,k_raygen").split(",")):
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d[k].items()})
PY
done
