mkdir -p gpurun_out/r05
timeout 300 python tools/roulette_sweep.py one
timeout 600 python tools/mesh_path_check.py small 2>&1 | grep -v queue | head -6
timeout 2700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mesh.py tests/test_gpu_fuzz.py tests/test_gpu_jit.py -m gpu -x -q 2>&1 | tail -4
