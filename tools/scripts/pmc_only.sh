cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rec; R=$GRAFT_REPO_ROOT/gpurun_out/rec
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/probe_traffic.py 416 > $R/pmc_f.log 2>&1; echo "pmc_f rc=$?" >> $R/pmc_f.log
python $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmc_f $R/pmc_fetch.json > $R/pmc_fetch.txt 2>&1; rm -rf /tmp/pmc_f
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/probe_traffic.py 416 > $R/pmc_w.log 2>&1; echo "pmc_w rc=$?" >> $R/pmc_w.log
python $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmc_w $R/pmc_write.json > $R/pmc_write.txt 2>&1; rm -rf /tmp/pmc_w
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/probe_traffic.py 416 > $R/traffic_plain.log 2>&1
timeout -k 10 400 python bench.py > $R/bench.json 2> $R/bench.err
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ladder --no-shard --cpu-baseline-seconds 0 > $R/bench_line_under_rocprof.json 2> $R/rocprof.err; echo "rocprof rc=$?" >> $R/rocprof.err
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} $R/kernel_stats.csv \;
rm -rf /tmp/prof_b
