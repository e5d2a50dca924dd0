# the part of tools/evidence.sh that the shade launches of mesh scenes touch (run after a change to k_shade<TAIL>)
set -u
T=r05
E=$PWD/gpurun_out/evidence_$T
mkdir -p "$E"
bash tools/profile.sh ${T}_mesh mesh160x160:512x512x64:d8:fwdbwd --scene mesh160x160 > "$E/prof2.log" 2>&1
bash tools/profile.sh ${T}_config4 mesh160x160:1024x1024x32:d8:fwdbwd --config 4 >> "$E/prof2.log" 2>&1
bash tools/profile.sh ${T}_unbiased_mesh mesh160x160:512x512x64:d8:unbiased  --scene mesh160x160 --unbiased >> "$E/prof2.log" 2>&1
python3 tools/merge_traffic.py profiles/traffic.json gpurun_out/prof_${T}_mesh/traffic.json gpurun_out/prof_${T}_config4/traffic.json gpurun_out/prof_${T}_unbiased_mesh/traffic.json >> "$E/prof2.log" 2>&1
cp profiles/traffic.json "$E/traffic_merged.json"
python3 bench.py --config 4 > "$E/bench_config4_per_gpu_share.json" 2>> "$E/bench2.err"
python3 bench.py --config 4 --per-face > "$E/bench_config4_per_face.json" 2>> "$E/bench2.err"
python3 bench.py --config 4 --spp 256 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-views > "$E/bench_config4_full_size_one_gpu.json" 2>> "$E/bench2.err"
python3 bench.py --scene mesh160x160 > "$E/bench_mesh160x160_512x512x64.json" 2>> "$E/bench2.err"
python3 bench.py --scene mesh160x160 --unbiased --no-extra-views --steps 5 --warmup 2 > "$E/bench_unbiased_mesh160x160.json" 2>> "$E/bench2.err"
python3 tools/mesh_scale.py > "$E/mesh_scale.txt" 2>&1
python3 tools/walk_diag.py - mesh160x160 64 > "$E/walk_by_depth.txt" 2>&1
python3 bench.py > "$E/bench_after_mesh_changes.json" 2>> "$E/bench2.err"
for t in ${T}_mesh ${T}_config4 ${T}_unbiased_mesh; do
  P=gpurun_out/prof_$t
  cp $P/summary.txt "$E/${t}_rocprofv3_summary.txt"
  cp $P/traffic.json "$E/${t}_traffic.json"
  cp $P/summary.json "$E/${t}_rocprofv3_summary.json"
  cp $P/bench_trace.json "$E/${t}_bench_under_rocprof.json"
  cp $(ls $P/trace/*/*kernel_stats.csv | head -1) "$E/${t}_kernel_stats.csv"
done
tail -2 "$E/bench2.err"; cut -c1-300 "$E/bench_unbiased_mesh160x160.json"; cut -c1-200 "$E/bench_after_mesh_changes.json"
