#!/bin/bash
# Regenerate the rocprofv3 summaries and experiment tables kept under profiles/ (run on the GPU box:
#   gpurun -- 'bash scripts/make_profiles.sh r04').  Kernel traces and PMC passes are separate runs (a --pmc run never carries a trace
# domain); everything lands in gpurun_out/profiles/ and is copied into profiles/ by hand.
set -u
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline"
STEP="python $ROOT/scripts/prof_step.py"
# 0. the bench lines themselves (fp32 = BASELINE config 2, bf16 = config 3), builder-run; the driver's 20-step form; the B = 8 regime of round 3
python $ROOT/bench.py --steps 160 > $OUT/${R}_bench_fp32.json 2> $OUT/bench_fp32.err
$B --steps 160 --precision bf16 > $OUT/${R}_bench_bf16.json 2> $OUT/bench_bf16.err
python $ROOT/bench.py --steps 20 --warmup 4 > $OUT/${R}_bench_driver_style.json 2> $OUT/bench_driver.err
$B --steps 160 --coalesce 1 --streams 20 > $OUT/${R}_bench_fp32_uncoalesced_20streams.json 2> /dev/null
# 1. the default bench command under the kernel trace (the profiler serialises the streams)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- $B --steps 60 --min-seconds 1 > $OUT/bench.log 2>&1
cp $OUT/bench/b_kernel_stats.csv $OUT/${R}_kernel_stats_bench_default.csv
# 2. one call at a time, eager launches: per-kernel durations at 240 clouds per call (the executor's call) and at 8 (a caller that cannot batch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e240 -o e -- $STEP 240 10 > $OUT/e240.log 2>&1
cp $OUT/e240/e_kernel_stats.csv $OUT/${R}_kernel_stats_eager_240clouds.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e240b -o e -- $STEP 240 10 bf16 > $OUT/e240b.log 2>&1
cp $OUT/e240b/e_kernel_stats.csv $OUT/${R}_kernel_stats_eager_240clouds_bf16.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e8 -o e -- $STEP 8 40 > $OUT/e8.log 2>&1
cp $OUT/e8/e_kernel_stats.csv $OUT/${R}_kernel_stats_eager_8clouds.csv
# 3. HBM traffic counters at 240 clouds per call, one pass per counter
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- $STEP 240 4 > $OUT/pmc_$C.log 2>&1
done
python $ROOT/scripts/pmc_to_profile.py $OUT/${R}_pmc_hbm_traffic.csv $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv')
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmcb_$C -o p -- $STEP 240 4 bf16 > $OUT/pmcb_$C.log 2>&1
done
python $ROOT/scripts/pmc_to_profile.py $OUT/${R}_pmc_hbm_traffic_bf16.csv $(find $OUT/pmcb_FETCH_SIZE $OUT/pmcb_WRITE_SIZE -name '*counter_collection.csv')
# 4. SQ counters (matrix-pipe busy, issue stalls, LDS conflicts, instruction mix) at 240 clouds per call, fp32 and bf16
PMC_CMD="$STEP 240 4" bash $ROOT/scripts/make_pmc_sq.sh $R > $OUT/pmc_sq.log 2>&1
PMC_SUFFIX=_bf16 PMC_CMD="$STEP 240 4 bf16" bash $ROOT/scripts/make_pmc_sq.sh $R > $OUT/pmc_sq_bf16.log 2>&1
# 5. BASELINE config 4 (8 clips x 30 frames) whole model
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4 -o m -- python $ROOT/scripts/time_model.py 8 30 8192 3 > $OUT/cfg4.log 2>&1
cp $OUT/cfg4/m_kernel_stats.csv $OUT/${R}_kernel_stats_cfg4_model.csv
python $ROOT/scripts/time_model.py 8 30 8192 3 > $OUT/${R}_cfg4_model_time.txt 2>&1
# 6. BASELINE config 5 stress (ball query + grouped MLP at N = 32768) and the sampling of those clouds
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5 -o s -- python $ROOT/scripts/stress_cfg5.py > $OUT/cfg5.log 2>&1
cp $OUT/cfg5/s_kernel_stats.csv $OUT/${R}_kernel_stats_cfg5_stress.csv
python $ROOT/scripts/time_fps_big.py 2>&1 | grep -v amdgpu > $OUT/${R}_fps_large_clouds.txt
# 6b. config 5's counters (BASELINE: "HBM-roofline run with rocprof counters"): HBM traffic and SQ counters of the stress launches
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc5_$C -o p -- python $ROOT/scripts/stress_cfg5.py > $OUT/pmc5_$C.log 2>&1
done
python $ROOT/scripts/pmc_to_profile.py $OUT/${R}_pmc_hbm_traffic_cfg5.csv $(find $OUT/pmc5_FETCH_SIZE $OUT/pmc5_WRITE_SIZE -name '*counter_collection.csv')
PMC_SUFFIX=_cfg5 PMC_CMD="python $ROOT/scripts/stress_cfg5.py" bash $ROOT/scripts/make_pmc_sq.sh $R > $OUT/pmc_sq_cfg5.log 2>&1
rm -rf $OUT/pmc5_FETCH_SIZE $OUT/pmc5_WRITE_SIZE
# 6c. lbs() by batch size: the matrix-pipe route (round 5) against round 4's one-launch kernel and the three-launch route
python $ROOT/scripts/time_lbs.py 1 8 16 30 240 2>&1 | grep "lbs()" > $OUT/${R}_lbs_by_batch.txt
# 7. frames/s over (steps coalesced per call) x (calls in flight)
cd $ROOT
{
  echo "# bench.py --no-cpu-baseline --min-seconds 1.5 --coalesce k --streams ns: frames/s (us per B=8 step) | one call alone, ms"
  for cfg in "1 1" "1 8" "1 16" "1 20" "2 12" "4 8" "8 4" "8 6" "15 2" "15 4" "30 1" "30 2" "30 3" "30 4" "60 2"; do
    set -- $cfg
    python bench.py --no-cpu-baseline --min-seconds 1.5 --coalesce $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('k=%3d clouds/call=%4d calls in flight=%2d clouds in flight=%4d : %7.0f frames/s (%6.1f us/step) | %7.3f ms' % ($1, 8*$1, $2, 8*$1*$2, d['value'], d['ms_per_step']*1e3, d['latency_ms_one_call']))"
  done
} > $OUT/${R}_coalesce_by_streams.txt
# 8. which launches of a call run concurrently with which (recorded C-ABI calls replayed alone and in pairs on two streams); per-launch times in bf16
timeout 600 python scripts/exp_overlap.py 240 2>/dev/null > $OUT/${R}_overlap_pairs_240clouds.txt
PAIRS="fps_gather_grid:mlp_chain_bf16,fps_gather_grid:three_nn" timeout 600 python scripts/exp_overlap.py 240 bf16 2>/dev/null > $OUT/${R}_launches_240clouds_bf16.txt
python scripts/time_gemm.py 2>&1 | grep -v amdgpu > $OUT/${R}_gemm_shapes.txt
# 8b. gemm_tile.hip: phase stamps (debug build) and the SUSTAINED rate next to hipBLASLt's, with the shader clock (time_gemm.py's 7 ms bursts from an
#     idle chip read ~12 % lower than a launch inside a running call)
[ -f $ROOT/garment4d_amd/lib/libg4d_hip_dbg.so ] && {
  echo "# gemm_tile.hip (g4d_linear_f32 at the wide FP level's shapes): where a workgroup's cycles go, and the sustained rate next to hipBLASLt's"
  echo "# scripts/dbg_gemm_phases.py on garment4d_amd/lib/libg4d_hip_dbg.so (make dbg: -DG4D_GEMM_DEBUG, cycle stamps of wave 0 of every workgroup; the stamps cost ~10 % themselves),"
  echo "# then scripts/exp_clock_gemm.py (one kernel back to back for 2 s, shader clock from scripts/micro/clockprobe.hip, board power from rocm-smi)"
  G4D_LIB_PATH=$ROOT/garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_gemm_phases.py 2>&1 | grep -v amdgpu
  G4D_LIB_PATH=$ROOT/garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_gemm_phases.py 61440 512 256 2>&1 | grep -v amdgpu
  python scripts/exp_clock_gemm.py 2>&1 | grep -v amdgpu
} > $OUT/${R}_gemm_tile_phases.txt
python scripts/exp_clock.py 2>&1 | grep -v amdgpu > $OUT/${R}_clock_and_power_by_regime.txt
rm -rf $OUT/bench $OUT/e240 $OUT/e240b $OUT/e8 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcb_FETCH_SIZE $OUT/pmcb_WRITE_SIZE $OUT/cfg4 $OUT/cfg5
ls -la $OUT
