#!/bin/bash
# SQ counter passes over one-batch-at-a-time eager launches of the cfg2 step (run on the GPU box):
#   gpurun -- 'bash scripts/make_pmc_sq.sh r03 [extra bench.py flags]'   ->  gpurun_out/profiles/<R>_pmc_sq[_suffix].csv
# One rocprofv3 --pmc run per counter group (<= 8 SQ counters per pass, GRBM_GUI_ACTIVE rides along in its own block); no trace domain
# is combined with --pmc.
set -u
R=${1:-r03}; shift || true
SUF=${PMC_SUFFIX:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC_CMD overrides the profiled command (e.g. "python $ROOT/scripts/prof_step.py 240 4": the step on 240 clouds per call)
B=${PMC_CMD:-"python $ROOT/bench.py --no-cpu-baseline --steps 32 --streams 1 --no-graph --min-seconds 0 $*"}
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
G3="SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rm -rf $OUT/sq$i
  rocprofv3 --pmc $G --output-format csv -d $OUT/sq$i -o p -- $B > $OUT/sq$i.log 2>&1 || echo "pass $i failed (see $OUT/sq$i.log)"
done
python $ROOT/scripts/pmc_sq_to_profile.py $OUT/${R}_pmc_sq${SUF}.csv $(find $OUT/sq1 $OUT/sq2 $OUT/sq3 -name '*counter_collection.csv' 2>/dev/null)
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
head -40 $OUT/${R}_pmc_sq${SUF}.csv | cut -c1-250
