#!/bin/bash
# round-3 experiment A: XCD-contiguous row blocks in the chain kernel, weight-streaming diagnostic
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expA; mkdir -p $O
python -m pytest tests/test_fused_gpu.py tests/test_encoder_gpu.py tests/test_parity_fullsize_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 160"
$B > $O/bench_xcd1.json 2> $O/bench_xcd1.err
G4D_CHAIN_XCD=0 $B > $O/bench_xcd0.json 2> $O/bench_xcd0.err
G4D_CHAIN_DBG_WSAME=1 $B > $O/bench_wsame.json 2> $O/bench_wsame.err
python scripts/exp_saturated_cost.py > $O/sat_xcd1.txt 2>&1
G4D_CHAIN_XCD=0 python scripts/exp_saturated_cost.py > $O/sat_xcd0.txt 2>&1
G4D_CHAIN_DBG_WSAME=1 python scripts/exp_saturated_cost.py > $O/sat_wsame.txt 2>&1
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['latency_ms_single_stream'], d['roofline_mfma']['avg_launch_us'])"; done
