#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expK; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-10s %7.1f us/step %6.0f frames/s' % ('$name', d['ms_per_step']*1e3, d['value']))"; }
run base X=1
run nostore G4D_CHAIN_DBG_WSAME=2
run nogather G4D_CHAIN_DBG_WSAME=4
run neither G4D_CHAIN_DBG_WSAME=6
KNOCK=g4d_mlp_chain_table_cells_f32 python scripts/exp_knockout.py --steps 160 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knock fp1  %7.1f us/step' % (d['ms_per_step']*1e3))"
run base2 X=1
