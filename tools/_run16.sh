export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05/pmc_regen
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -- python3 tools/roulette_sweep.py one > $OUT/set$i.out 2> $OUT/set$i.err
  python3 - $OUT/set$i <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:50]
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in d:
    if "k_path<" in k:
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d[k].items()}, "n", len(next(iter(d[k].values()))))
PY
done
find $OUT -name "*.csv" -size +1M -delete
