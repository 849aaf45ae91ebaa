#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expH; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_parity_fullsize_gpu.py tests/test_cuda_pin.py tests/test_bench_gpu.py -x -q -m gpu -s --durations=8 > $O/pytest.log 2>&1; grep -E "parity\]|passed|failed|Error|slowest|s call" $O/pytest.log | tail -60
