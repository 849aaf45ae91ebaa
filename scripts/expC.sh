#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expC; mkdir -p $O
python -m pytest tests/test_fused_gpu.py tests/test_encoder_gpu.py tests/test_bench_gpu.py tests/test_lbs_gcn_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), d['latency_ms_single_stream'])"; }
run g1 X=1
run g0 G4D_LAUNCH_GROUPS=0
run g1b X=1
run g0b G4D_LAUNCH_GROUPS=0
B="python bench.py --no-cpu-baseline --steps 160 --streams 20"; run g1s20 X=1
run g0s20 G4D_LAUNCH_GROUPS=0
