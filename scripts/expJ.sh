#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expJ; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_encoder_gpu.py tests/test_parity_fullsize_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), d['latency_ms_single_stream'])"; }
run cells1 X=1
run cells0 G4D_FP_CELLS=0
run cells1b X=1
run cells0b G4D_FP_CELLS=0
