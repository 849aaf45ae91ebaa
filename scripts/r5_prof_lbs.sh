cd /tmp && export TMPDIR=/tmp
for B in 8 240; do rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_lbs$B; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lbs$B -o lbs -- python $GRAFT_REPO_ROOT/scripts/time_lbs.py $B 2>&1 | grep "lbs()"; done
cd $GRAFT_REPO_ROOT; for B in 8 240; do python scripts/kstats.py $(ls gpurun_out/prof_lbs$B/*kernel_stats.csv) 2>&1 | grep "lbs_one\|lbs_mfma\|lbs_frame"; done
