#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expN; mkdir -p $O
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-10s %7.1f us/step %6.0f frames/s lat %.3f' % ('$name', d['ms_per_step']*1e3, d['value'], d['latency_ms_single_stream']))"; }
A=$PWD/garment4d_amd/lib/libg4d_agpr.so
for i in 1 2 3; do
B="timeout 300 python bench.py --no-cpu-baseline --steps 160"; run vgpr$i X=1; run agpr$i G4D_LIB_PATH=$A
done
B="timeout 300 python bench.py --no-cpu-baseline --steps 160 --precision bf16"; run vgpr_bf16 X=1; run agpr_bf16 G4D_LIB_PATH=$A; run vgpr_bf16b X=1; run agpr_bf16b G4D_LIB_PATH=$A
B="timeout 300 python scripts/time_model.py 8 30 8192 3"
python scripts/time_model.py 8 30 8192 3 2>&1 | grep -v amdgpu | tail -3
G4D_LIB_PATH=$A python scripts/time_model.py 8 30 8192 3 2>&1 | grep -v amdgpu | tail -3
