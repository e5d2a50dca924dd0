#!/bin/bash
# Where does k_intersect_mesh (the BVH walk) wait: the vector-memory pipeline by its own counters -- TA (address), TCP (L1: tag
# lookups, pending stalls), TD (data return) -- beside the SQ's view.  Run through gpurun.  Usage: tools/pmc_walk.sh [bench args]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_walk
mkdir -p "$OUT"
# (a counter name the tool does not know makes it abort and HANG: every pass runs under `timeout`; names from rocprofv3 --list-avail)
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  d=$OUT/set$i
  if [ -n "$PMC_SETS" ] && ! echo " $PMC_SETS " | grep -q " $i "; then continue; fi
  timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$d" -- python3 bench.py --steps 1 --warmup 0 --preheat-ms 0 --no-cpu-baseline --no-extra-views "$@" > /dev/null 2> "$d.err" || { echo "set $i failed:" >&2; tail -3 "$d.err" >&2; continue; }
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
    if any(x in k for x in os.environ.get("PMC_KERNELS", "k_intersect_mesh").split(",")):
        print(k, "launches", len(next(iter(d[k].values()))), {c: "%.4g" % (sum(v) / len(v)) for c, v in d[k].items()})
PY
done
