#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expG; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), d['latency_ms_single_stream'], d['roofline']['avg_launch_us'])"; }
run multi X=1
run single G4D_FPS_MULTI=1
run multi2 X=1
run single2 G4D_FPS_MULTI=1
B="timeout 300 python bench.py --no-cpu-baseline --steps 160 --streams 20"; run multi_s20 X=1
run single_s20 G4D_FPS_MULTI=1
