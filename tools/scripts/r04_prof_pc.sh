# kernel durations of the preconditioner set-up (two-launch and one-launch panel steps)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04pc; R=$GRAFT_REPO_ROOT/gpurun_out/r04pc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pc -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pc_fused.py 416 > $R/probe.log 2>&1; echo "rc=$?" >> $R/probe.log
find /tmp/prof_pc -name "*kernel_stats.csv" -exec cp {} $R/kernel_stats.csv \;
rm -rf /tmp/prof_pc
tail -8 $R/probe.log; head -12 $R/kernel_stats.csv | cut -c1-160
