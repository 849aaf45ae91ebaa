#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expE; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_contraction.py tests/test_fullsize_gpu.py tests/test_encoder_gpu.py tests/test_parity_fullsize_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 120 python scripts/time_fps.py > $O/time_fps_multi.txt 2>&1; tail -12 $O/time_fps_multi.txt
G4D_FPS_MULTI=1 timeout 120 python scripts/time_fps.py > $O/time_fps_single.txt 2>&1; tail -12 $O/time_fps_single.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), d['latency_ms_single_stream'], d['roofline']['avg_launch_us'])"; }
run multi X=1
run single G4D_FPS_MULTI=1
B="timeout 300 python bench.py --no-cpu-baseline --steps 160 --streams 20"; run multi_s20 X=1
B="timeout 300 python bench.py --no-cpu-baseline --steps 160 --streams 12"; run multi_s12 X=1
