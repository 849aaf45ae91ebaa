#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expF; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_contraction.py tests/test_fullsize_gpu.py -x -q -m gpu -k "fps or FPS or sampling or cfg2" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 120 python scripts/dbg_fps_multi.py 2>&1 | grep -v amdgpu | head -4
G4D_FPS_BUCKET_W=16 G4D_FPS_DEAL=8 timeout 120 python scripts/time_fps.py run 2>&1 | grep -v amdgpu
for k in 2 3 4 6; do echo kcap $k; G4D_FPS_KCAP=$k G4D_FPS_BUCKET_W=16 G4D_FPS_DEAL=8 timeout 120 python scripts/time_fps.py run 2>&1 | grep -v amdgpu; done
G4D_FPS_MULTI=1 G4D_FPS_BUCKET_W=16 G4D_FPS_DEAL=8 timeout 120 python scripts/time_fps.py run 2>&1 | grep -v amdgpu
