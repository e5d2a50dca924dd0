#!/bin/bash
# Run on the GPU box (through gpurun): regenerate every file of one round's evidence from the SAME binary.
# Usage: tools/evidence.sh <tag>      outputs under gpurun_out/evidence_<tag>/ (copy into profiles/ with tools/collect_evidence.sh)
set -u
T=${1:-r03}
E=$PWD/gpurun_out/evidence_$T
mkdir -p "$E"
bash tools/profile.sh $T cornell:512x512x64:d8:fwdbwd > "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_streaming cornell:512x512x64:d8:fwdbwd --bounces-per-launch 1 >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_mesh mesh160x160:512x512x64:d8:fwdbwd --scene mesh160x160 >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_config5 cornell_specular:2048x2048x128:d16:fwdbwd --config 5 >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_config4 mesh160x160:1024x1024x32:d8:fwdbwd --config 4 >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_roulette cornell:512x512x64:rr0.5b1:fwdbwd --absorb 0.5 --min-bounces 1 >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_unbiased cornell:512x512x64:d8:unbiased --unbiased >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_fwd cornell:512x512x64:d8:fwd --config 2 >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_unbiased_mesh mesh160x160:512x512x64:d8:unbiased  --scene mesh160x160 --unbiased >> "$E/prof.log" 2>&1
bash tools/profile.sh ${T}_many_params cornell_shapes:512x512x64:d8:fwdbwd --scene cornell_shapes >> "$E/prof.log" 2>&1
# bench.py quotes PMC numbers only for the workloads profiled above (one entry each)
python3 tools/merge_traffic.py profiles/traffic.json gpurun_out/prof_$T/traffic.json gpurun_out/prof_${T}_mesh/traffic.json gpurun_out/prof_${T}_config5/traffic.json gpurun_out/prof_${T}_config4/traffic.json gpurun_out/prof_${T}_roulette/traffic.json gpurun_out/prof_${T}_unbiased/traffic.json gpurun_out/prof_${T}_unbiased_mesh/traffic.json gpurun_out/prof_${T}_fwd/traffic.json gpurun_out/prof_${T}_many_params/traffic.json >> "$E/prof.log" 2>&1
cp profiles/traffic.json "$E/traffic_merged.json"
python3 tools/parity_report.py --big > "$E/parity_report.txt" 2> "$E/parity_report.err"
# (--store-n1: profiles/n1_reference.json is re-stored at this binary, so the first N > 1 run compares against today's value)
python3 bench.py --store-n1 > "$E/bench.json" 2> "$E/bench.err"
cp profiles/n1_reference.json "$E/n1_reference.json"
# a node with more than one GPU: the scaling lines (bench.py starts its own ranks, one per GPU; weak scaling)
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)
for n in 2 4 8; do
  if [ "$NGPU" -ge "$n" ]; then
    python3 bench.py --gpus $n --no-cpu-baseline --no-extra-views 2>> "$E/bench.err" | grep "^{" > "$E/scale_config3_${n}gpus.json"
    python3 bench.py --gpus $n --config 4 --no-cpu-baseline --no-extra-views 2>> "$E/bench.err" | grep "^{" > "$E/scale_config4_${n}gpus.json"
    python3 bench.py --gpus $n --config 5 --no-cpu-baseline --no-extra-views 2>> "$E/bench.err" | grep "^{" > "$E/scale_config5_${n}gpus.json"
  fi
done
(python3 tools/fit_albedo.py --quiet; python3 tools/fit_albedo.py --quiet --async) > "$E/fit_albedo.txt" 2>&1
python3 bench.py --config 2 --no-cpu-baseline > "$E/bench_config2_fwd_only.json" 2>> "$E/bench.err"
python3 bench.py --config 4 > "$E/bench_config4_per_gpu_share.json" 2>> "$E/bench.err"
python3 bench.py --config 5 > "$E/bench_config5_per_gpu_share.json" 2>> "$E/bench.err"
python3 bench.py --config 4 --spp 256 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-views > "$E/bench_config4_full_size_one_gpu.json" 2>> "$E/bench.err"
python3 bench.py --config 5 --spp 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-views > "$E/bench_config5_full_size_one_gpu.json" 2>> "$E/bench.err"
python3 bench.py --scene mesh160x160 > "$E/bench_mesh160x160_512x512x64.json" 2>> "$E/bench.err"
python3 bench.py --absorb 0.5 --min-bounces 1 > "$E/bench_roulette_b1_p0.5.json" 2>> "$E/bench.err"
python3 bench.py --unbiased --no-extra-views > "$E/bench_unbiased.json" 2>> "$E/bench.err"
python3 bench.py --gpus 2 --dist-backend gloo --same-gpu --no-cpu-baseline --no-extra-views 2>> "$E/bench.err" | grep "^{" > "$E/bench_2ranks_same_gpu_plumbing.json"
python3 tools/mesh_scale.py > "$E/mesh_scale.txt" 2>&1
python3 bench.py --scene mesh160x160 --unbiased --no-extra-views --steps 5 --warmup 2 > "$E/bench_unbiased_mesh160x160.json" 2>> "$E/bench.err"
python3 tools/async_timing.py > "$E/async_host_buffers.txt" 2>&1
timeout 900 python3 tools/fuzz_reference.py 400 11 > "$E/fuzz_vs_reference.txt" 2>&1
timeout 900 python3 tools/fuzz_overlap.py 300 1 2>&1 | grep -v amdgpu > "$E/fuzz_overlap.txt"
(echo "## frames overlapping (default)"; python3 tools/one_ctx_frames.py 2>&1 | grep -v amdgpu; echo "## DRT_HIP_OVERLAP_FRAMES=0"; DRT_HIP_OVERLAP_FRAMES=0 python3 tools/one_ctx_frames.py 2>&1 | grep -v amdgpu) > "$E/one_ctx_frames.txt"
python3 tools/two_frames.py cornell 2>&1 | grep -v amdgpu > "$E/two_frames.txt"
(python3 tools/jit_background.py random11; python3 tools/jit_background.py random5) 2>&1 | grep -v amdgpu > "$E/jit_background.txt"
python3 tools/small_frames.py 2>&1 | grep -v amdgpu > "$E/small_frames.txt"
# round 5: the synchronous call in its four forms; mesh scenes through both routes at small frame sizes; the roulette kernel's knobs;
# config 4 with an albedo parameter per face; the group context's own bench mode (here: two members on the one device)
python3 tools/sync_call.py 2>&1 | grep -v amdgpu > "$E/sync_call.txt"
python3 tools/mesh_path_check.py small 2>&1 | grep -v amdgpu > "$E/mesh_small_frames.txt"
python3 tools/mesh_path_check.py parity 2>&1 | grep -v amdgpu > "$E/mesh_path_parity.txt"
python3 tools/roulette_sweep.py 2>&1 | grep -v amdgpu > "$E/roulette_sweep.txt"
python3 tools/tail_check.py time 2>&1 | grep -v amdgpu > "$E/two_stage_shade_times.txt"
python3 bench.py --config 4 --per-face > "$E/bench_config4_per_face.json" 2>> "$E/bench.err"
python3 bench.py --single-process --gpus 2 --same-gpu 2>> "$E/bench.err" | grep "^{" > "$E/bench_group_2members_same_gpu_plumbing.json"
(timeout 300 python3 tools/allreduce_overlap.py; timeout 300 python3 tools/allreduce_overlap.py --torch-dist-eager; timeout 300 python3 tools/allreduce_overlap.py --torch-dist-eager --context-between) 2>&1 | grep "ms per frame\|initialised" > "$E/launch_order.txt"
(timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --no-cpu-baseline --no-extra-views 2>> "$E/bench.err" | grep "^{") > "$E/bench_1rank_under_torchrun.json"
# round 6: what the number of scene parameters costs (the general form of the one-launch kernels); many-parameter bench lines;
# the f64 route's kernel time
python3 tools/param_cliff.py 2>&1 | grep -v amdgpu > "$E/param_cliff.txt"
for sc in cornell_shapes params16 params32 params64; do python3 bench.py --scene $sc --no-extra-views --no-cpu-baseline > "$E/bench_$sc.json" 2>> "$E/bench.err"; done
python3 tools/f64_frames.py 10 2>&1 | grep -v amdgpu > "$E/f64_frames.txt"
python3 tools/walk_diag.py - mesh160x160 64 > "$E/walk_by_depth.txt" 2>&1
[ -f build/lib_stats.so ] && python3 tools/bvh_stats.py build/lib_stats.so > "$E/bvh_stats.txt" 2>&1
for t in $T ${T}_streaming ${T}_mesh ${T}_config5 ${T}_config4 ${T}_roulette ${T}_unbiased ${T}_unbiased_mesh ${T}_fwd ${T}_many_params; do
  P=gpurun_out/prof_$t
  cp $P/summary.txt "$E/${t}_rocprofv3_summary.txt"
  cp $P/traffic.json "$E/${t}_traffic.json"
  cp $P/summary.json "$E/${t}_rocprofv3_summary.json"
  cp $P/bench_trace.json "$E/${t}_bench_under_rocprof.json"
  cp $(ls $P/trace/*/*kernel_stats.csv | head -1) "$E/${t}_kernel_stats.csv"
done
tail -3 "$E/bench.err"
cat "$E/bench.json" | cut -c1-400
