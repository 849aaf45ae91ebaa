#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expL; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_encoder_gpu.py tests/test_parity_fullsize_gpu.py tests/test_lbs_gcn_gpu.py tests/test_refine_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --steps 160"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-10s %7.1f us/step %6.0f frames/s lat %.3f mfma %.3f %.1fus' % ('$name', d['ms_per_step']*1e3, d['value'], d['latency_ms_single_stream'], d['roofline_mfma']['frac'], d['roofline_mfma']['avg_launch_us']))"; }
run base X=1
run mt2 G4D_CHAIN_MT=2
run mt1 G4D_CHAIN_MT=1
run wide1024 G4D_CHAIN_MT2_MIN_WAVES_WIDE=1024
run mt2min1024 G4D_CHAIN_MT2_MIN_WAVES=1024
run base2 X=1
B="timeout 300 python bench.py --no-cpu-baseline --steps 160 --streams 20"; run s20 X=1
B="timeout 300 python bench.py --no-cpu-baseline --steps 160 --precision bf16"; run bf16 X=1
