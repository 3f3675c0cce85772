# round-3 records of the final build: rocprofv3 kernel stats of the bench command (the one-launch frozen year is the dominant
# kernel now), HBM traffic of that kernel (separate --pmc passes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rec4; R=$GRAFT_REPO_ROOT/gpurun_out/rec4
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --no-ladder --no-shard --no-mix --no-shard3 --cpu-baseline-seconds 0 > $R/bench_line_under_rocprof.json 2> $R/rocprof.err; echo "rocprof rc=$?" >> $R/rocprof.err
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} $R/kernel_stats.csv \;
rm -rf /tmp/prof_b
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_traffic.py 416 > $R/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?" >> $R/pmc_$tag.log
  python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmc_$tag $R/pmc_$tag.json > $R/pmc_$tag.txt 2>&1; rm -rf /tmp/pmc_$tag
done
ls -la $R
