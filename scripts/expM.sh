#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expM; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-10s %7.1f us/step %6.0f frames/s lat %.3f' % ('$name', d['ms_per_step']*1e3, d['value'], d['latency_ms_single_stream']))"; }
run base X=1
run fpwide G4D_FP_WIDE_FUSED=1
run base2 X=1
run fpwide2 G4D_FP_WIDE_FUSED=1
run nocells G4D_FP_CELLS=0
