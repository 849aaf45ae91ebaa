#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expB; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), d['latency_ms_single_stream'])"; }
run base X=1
run mt1 G4D_CHAIN_MT=1
run mt2 G4D_CHAIN_MT=2
B="python bench.py --no-cpu-baseline --steps 160 --streams 12"; run s12 X=1
B="python bench.py --no-cpu-baseline --steps 160 --streams 20"; run s20 X=1
B="python bench.py --no-cpu-baseline --steps 160 --streams 24"; run s24 X=1
B="python bench.py --no-cpu-baseline --steps 160 --streams 32"; GPU_MAX_HW_QUEUES=64 run s32 X=1
