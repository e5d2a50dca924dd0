#!/bin/bash
# Here (after gpurun merged gpurun_out/ back): copy one round's evidence into profiles/.  Usage: tools/collect_evidence.sh <tag>
set -eu
T=${1:-r03}
E=gpurun_out/evidence_$T
for f in "$E"/${T}*; do cp "$f" profiles/; done                   # the per-profile files already carry the tag
for f in parity_report.txt mesh_scale.txt bench.json bench_config2_fwd_only.json bench_config4_per_gpu_share.json \
         bench_config5_per_gpu_share.json bench_config4_full_size_one_gpu.json bench_config5_full_size_one_gpu.json \
         bench_mesh160x160_512x512x64.json bench_roulette_b1_p0.5.json bench_unbiased.json bench_2ranks_same_gpu_plumbing.json \
         bench_unbiased_mesh160x160.json fit_albedo.txt scale_config3_2gpus.json scale_config3_4gpus.json scale_config3_8gpus.json scale_config4_2gpus.json scale_config4_4gpus.json scale_config4_8gpus.json scale_config5_2gpus.json scale_config5_4gpus.json scale_config5_8gpus.json fuzz_vs_reference.txt fuzz_overlap.txt one_ctx_frames.txt two_frames.txt async_host_buffers.txt walk_by_depth.txt bvh_stats.txt jit_background.txt small_frames.txt launch_order.txt bench_1rank_under_torchrun.json sync_call.txt mesh_small_frames.txt mesh_path_parity.txt roulette_sweep.txt bench_config4_per_face.json bench_group_2members_same_gpu_plumbing.json two_stage_shade_times.txt param_cliff.txt bench_cornell_shapes.json bench_params16.json bench_params32.json bench_params64.json f64_frames.txt; do
  [ -f "$E/$f" ] || continue
  cp "$E/$f" "profiles/${T}_$f"
done
cp "$E/traffic_merged.json" profiles/traffic.json
[ -f "$E/n1_reference.json" ] && cp "$E/n1_reference.json" profiles/n1_reference.json
ls -la profiles | grep "${T}_" | wc -l
