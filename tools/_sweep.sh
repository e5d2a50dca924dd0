run() { echo "== $*"; env "$@" timeout 300 python tools/mesh_scale.py 2>&1 | tail -2; }
run DRT_HIP_K2_PATHS_PER_BLOCK=1024 DRT_HIP_K2_MAX_BLOCKS_PER_CU=64
run DRT_HIP_K2_PATHS_PER_BLOCK=512 DRT_HIP_K2_MAX_BLOCKS_PER_CU=128
for r in 8 24 32; do run DRT_HIP_BVH_REFILL=$r; done
for d in 16 24 40 48; do run DRT_HIP_BVH_DESCEND_MIN=$d; done
run DRT_HIP_BVH_REFILL=8 DRT_HIP_BVH_DESCEND_MIN=24
