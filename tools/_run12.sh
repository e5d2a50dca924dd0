mkdir -p gpurun_out/r05
timeout 2700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05/gputests_parity.txt 2>&1
tail -12 gpurun_out/r05/gputests_parity.txt
