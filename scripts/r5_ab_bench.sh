# A/B of one environment switch through bench.py (both arms in one gpurun call: box-to-box spread is +-4 %)
# usage: bash scripts/r5_ab_bench.sh VAR "v1 v2 ..." [extra bench args]
var=$1; vals=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2; do for v in $vals; do
  echo -n "$var=$v: " | tee -a gpurun_out/r5_ab.log
  env $var=$v python bench.py --no-cpu-baseline --min-seconds 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('latency_ms_single_stream'))" | tee -a gpurun_out/r5_ab.log
done; done
