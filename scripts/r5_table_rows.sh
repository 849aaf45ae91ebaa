# the per-launch table of bench.py for a list of environment settings (one gpurun call; rows whose entry matches $PAT)
# usage: bash scripts/r5_table_rows.sh PAT VAR "v1 v2 ..."
pat=$1; var=$2; vals=$3
for v in $vals; do
  env $var=$v python bench.py --no-cpu-baseline --min-seconds 2 > gpurun_out/tr_$v.json 2>/dev/null
  echo "== $var=$v"; python scripts/show_bench.py gpurun_out/tr_$v.json | grep -E "frames/s|roofline_mfma_all|$pat"
done
