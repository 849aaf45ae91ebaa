#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expI; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fps" > $O/pytest_fps.log 2>&1; tail -4 $O/pytest_fps.log
timeout 600 python scripts/time_fps_big.py 2>&1 | grep -v amdgpu | tee $O/time_fps_big.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s --durations=5 -k "full_forward" > $O/pytest_model.log 2>&1; grep -E "parity\].*refinement|passed|failed|Error|s call" $O/pytest_model.log | tail -30
