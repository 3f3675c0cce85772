# round-4 records: rocprofv3 kernel stats of the bench command (does the profiled process end cleanly now that no launch is
# cooperative?), the slowdown probe (round-3 verdict weak 7), the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04rec1; R=$GRAFT_REPO_ROOT/gpurun_out/r04rec1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --no-ladder --no-shard --no-mix --no-shard3 --no-spinup --cpu-baseline-seconds 0 > $R/bench_line_under_rocprof.json 2> $R/rocprof.err; echo "rocprof rc=$?" >> $R/rocprof.err
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} $R/kernel_stats.csv \;
rm -rf /tmp/prof_b
tail -3 $R/rocprof.err
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/probe_slowdown.py 416 > $R/slowdown.log 2>&1; echo "rc=$?" >> $R/slowdown.log; cat $R/slowdown.log
timeout -k 10 900 python bench.py > $R/bench_default.json 2> $R/bench_default.err; echo "bench rc=$?" >> $R/bench_default.err; tail -25 $R/bench_default.err
