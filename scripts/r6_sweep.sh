#!/bin/bash
# frames/s over calls in flight / clouds per call on the current build
for cfg in "30 8" "30 10" "30 12" "30 6" "60 4" "60 6" "15 12" "15 16"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --no-dropin --min-seconds 2 --coalesce $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('k=%3d calls in flight=%2d : %7.0f frames/s (%6.1f us/step) | one call alone %7.3f ms | latency %6.3f ms' % ($1, $2, d['value'], d['ms_per_step']*1e3, d['latency_ms_one_call'], d['latency_ms_single_stream']))"
done
