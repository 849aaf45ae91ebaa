cd /tmp && export TMPDIR=/tmp
for B in 16 32 64 112 128 256 512; do rm -rf /tmp/pl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o lbs -- python $GRAFT_REPO_ROOT/scripts/time_lbs.py $B > /dev/null 2>&1; echo -n "B=$B "; python $GRAFT_REPO_ROOT/scripts/kstats.py $(ls /tmp/pl/*kernel_stats.csv) 2>&1 | grep "lbs_mfma" ; done
