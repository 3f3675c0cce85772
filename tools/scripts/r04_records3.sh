# round-4 records of the FINAL build (second session): the default bench line; rocprofv3 kernel stats of the bench command; HBM traffic and SQ
# counters of the dominant kernel (separate --pmc passes over tools/probe_traffic.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04rec3; R=$GRAFT_REPO_ROOT/gpurun_out/r04rec3
timeout -k 10 600 python bench.py > $R/bench_default.json 2> $R/bench_default.err; echo "bench rc=$?" >> $R/bench_default.err; tail -12 $R/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-ladder --no-shard --no-mix --no-shard3 --no-spinup --cpu-baseline-seconds 0 > $R/bench_line_under_rocprof.json 2> $R/rocprof.err; echo "rocprof rc=$?" >> $R/rocprof.err
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} $R/kernel_stats.csv \;
rm -rf /tmp/prof_b
tail -2 $R/rocprof.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  tag=$(echo $pass | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_traffic.py 416 > $R/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?" >> $R/pmc_$tag.log
  python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmc_$tag $R/pmc_$tag.json > $R/pmc_$tag.txt 2>&1; rm -rf /tmp/pmc_$tag
  tail -1 $R/pmc_$tag.log
done
python3 $GRAFT_REPO_ROOT/tools/make_traffic_summary.py $R/pmc_FETCH_SIZE.json $R/pmc_WRITE_SIZE.json $R/pmc_FETCH_SIZE.log $R/pmc_traffic_one_launch_416.json 416 | tail -12
ls -la $R
