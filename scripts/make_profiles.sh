#!/bin/bash
# Regenerate the rocprofv3 summaries and experiment tables kept under profiles/ (run on the GPU box:
#   gpurun -- 'bash scripts/make_profiles.sh r03').  Kernel traces and PMC passes are separate runs (a --pmc run never carries a trace
# domain); everything lands in gpurun_out/profiles/ and is copied into profiles/ by hand.
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline"
# 0. the bench lines themselves (fp32 = BASELINE config 2, bf16 = config 3), builder-run
python $ROOT/bench.py --steps 160 > $OUT/${R}_bench_fp32.json 2> $OUT/bench_fp32.err
$B --steps 160 --precision bf16 > $OUT/${R}_bench_bf16.json 2> $OUT/bench_bf16.err
# 1. the default bench command (16 batches in flight, hipGraph replay; the profiler serialises the streams)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- $B --steps 160 > $OUT/bench.log 2>&1
cp $OUT/bench/b_kernel_stats.csv $OUT/${R}_kernel_stats_bench_default.csv
# 2. one batch at a time, eager launches: per-kernel isolated durations
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -o e -- $B --steps 96 --streams 1 --no-graph --min-seconds 0 > $OUT/eager.log 2>&1
cp $OUT/eager/e_kernel_stats.csv $OUT/${R}_kernel_stats_eager_1stream.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager16 -o e -- $B --steps 96 --streams 1 --no-graph --min-seconds 0 --precision bf16 > $OUT/eager16.log 2>&1
cp $OUT/eager16/e_kernel_stats.csv $OUT/${R}_kernel_stats_eager_1stream_bf16.csv
# 3. HBM traffic counters, one pass per counter
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- $B --steps 32 --streams 1 --no-graph --min-seconds 0 > $OUT/pmc_$C.log 2>&1
done
python $ROOT/scripts/pmc_to_profile.py $OUT/${R}_pmc_hbm_traffic.csv $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv')
# 4. SQ counters (matrix-pipe busy, issue stalls, LDS conflicts, instruction mix), fp32 and bf16
bash $ROOT/scripts/make_pmc_sq.sh $R > $OUT/pmc_sq.log 2>&1
PMC_SUFFIX=_bf16 bash $ROOT/scripts/make_pmc_sq.sh $R --precision bf16 > $OUT/pmc_sq_bf16.log 2>&1
# 5. BASELINE config 4 (8 clips x 30 frames) whole model
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4 -o m -- python $ROOT/scripts/time_model.py 8 30 8192 3 > $OUT/cfg4.log 2>&1
cp $OUT/cfg4/m_kernel_stats.csv $OUT/${R}_kernel_stats_cfg4_model.csv
python $ROOT/scripts/time_model.py 8 30 8192 3 > $OUT/${R}_cfg4_model_time.txt 2>&1
# 6. BASELINE config 5 stress (ball query + grouped MLP at N = 32768) and the sampling of those clouds
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5 -o s -- python $ROOT/scripts/stress_cfg5.py > $OUT/cfg5.log 2>&1
cp $OUT/cfg5/s_kernel_stats.csv $OUT/${R}_kernel_stats_cfg5_stress.csv
python $ROOT/scripts/time_fps_big.py 2>&1 | grep -v amdgpu > $OUT/${R}_fps_large_clouds.txt
# 7. experiment tables: every launch alone vs 16 copies in flight; marginal cost of each launch in the 16-stream mix; dispatch cost
cd $ROOT
python scripts/exp_saturated_cost.py 2>&1 | grep -v amdgpu > $OUT/${R}_launch_cost_isolated_vs_16streams.txt
bash scripts/knockout_table.sh > $OUT/${R}_knockout_marginal_cost.txt 2>&1
python scripts/exp_dispatch.py 2>&1 | grep -v amdgpu > $OUT/${R}_dispatch_cost_by_streams.txt
python scripts/exp_launch_count.py 2>&1 | grep -v amdgpu >> $OUT/${R}_dispatch_cost_by_streams.txt
rm -rf $OUT/bench $OUT/eager $OUT/eager16 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/cfg4 $OUT/cfg5
ls -la $OUT
