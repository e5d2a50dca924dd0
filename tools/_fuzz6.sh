mkdir -p gpurun_out/r05
timeout 900 python3 tools/fuzz_mesh.py 16000 211 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260 > gpurun_out/r05/fuzz6_mesh_16000_summary.txt
timeout 700 python3 tools/fuzz_modes.py 14000 212 2>&1 | grep -v amdgpu | grep "beyond 2e-7\|FUZZ\|Error\|assert" | tail -6 | cut -c1-300 > gpurun_out/r05/fuzz6_routes_14000_summary.txt
timeout 500 python3 tools/fuzz_reference.py 10000 213 2>&1 | grep -v amdgpu | tail -1 | cut -c1-400 > gpurun_out/r05/fuzz6_vs_reference_10000_summary.txt
cat gpurun_out/r05/fuzz6_*_summary.txt
