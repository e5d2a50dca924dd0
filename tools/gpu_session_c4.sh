#!/bin/bash
set -u
O=gpurun_out/${1:-sc}; mkdir -p $O
(for sh in 0 1 2 3; do
 echo "## member stream shift $sh, default 5 blocks/CU"; DRT_HIP_MEMBER_STREAM_SHIFT=$sh python3 tools/overlap_shards.py 2>&1 | grep member | head -2
 echo "## member stream shift $sh, 4 blocks/CU"; DRT_HIP_MEMBER_STREAM_SHIFT=$sh DRT_HIP_MESH_BLOCKS_PER_CU=4 DRT_HIP_WALK_EXTRA_LDS=2048 python3 tools/overlap_shards.py 2>&1 | grep member | head -2
 echo "## member stream shift $sh, 3 blocks/CU"; DRT_HIP_MEMBER_STREAM_SHIFT=$sh DRT_HIP_MESH_BLOCKS_PER_CU=3 DRT_HIP_WALK_EXTRA_LDS=10240 python3 tools/overlap_shards.py 2>&1 | grep member | head -2
done) > $O/overlap.txt 2>&1
cat $O/overlap.txt
bash tools/profile.sh r04 cornell:512x512x64:d8:fwdbwd > $O/prof.log 2>&1
tail -28 $O/prof.log | head -12
