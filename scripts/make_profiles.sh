#!/bin/bash
# Regenerate the rocprofv3 summaries kept under profiles/ (run on the GPU box: gpurun -- 'bash scripts/make_profiles.sh r02').
# Kernel traces and PMC passes are separate runs (a --pmc run never carries a trace domain); outputs land in gpurun_out/profiles/.
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline"
# 1. the default bench command (16 batches in flight, hipGraph replay)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- $B --steps 160 > $OUT/bench.log 2>&1
cp $OUT/bench/b_kernel_stats.csv $OUT/${R}_kernel_stats_bench_default.csv
# 2. one batch at a time, eager launches: per-kernel isolated durations
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -o e -- $B --steps 96 --streams 1 --no-graph --min-seconds 0 > $OUT/eager.log 2>&1
cp $OUT/eager/e_kernel_stats.csv $OUT/${R}_kernel_stats_eager_1stream.csv
# 3. HBM traffic counters, one pass per counter
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- $B --steps 32 --streams 1 --no-graph --min-seconds 0 > $OUT/pmc_$C.log 2>&1
done
python $ROOT/scripts/pmc_to_profile.py $OUT/${R}_pmc_hbm_traffic.csv $OUT/pmc_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_WRITE_SIZE/p_counter_collection.csv
# 4. BASELINE config 4 (8 clips x 30 frames) whole model
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4 -o m -- python $ROOT/scripts/time_model.py 8 30 8192 3 > $OUT/cfg4.log 2>&1
cp $OUT/cfg4/m_kernel_stats.csv $OUT/${R}_kernel_stats_cfg4_model.csv
# 5. BASELINE config 5 stress (ball query at N = 131072)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5 -o s -- python $ROOT/scripts/stress_cfg5.py > $OUT/cfg5.log 2>&1
cp $OUT/cfg5/s_kernel_stats.csv $OUT/${R}_kernel_stats_cfg5_stress.csv
rm -rf $OUT/bench $OUT/eager $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/cfg4 $OUT/cfg5
ls -la $OUT
