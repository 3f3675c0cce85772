cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rec; R=$GRAFT_REPO_ROOT/gpurun_out/rec
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/probe_traffic.py 416 > $R/pmc_f.log 2>&1; echo "pmc_f rc=$?" >> $R/pmc_f.log
python $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmc_f $R/pmc_fetch.json > $R/pmc_fetch.txt 2>&1; rm -rf /tmp/pmc_f
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/probe_traffic.py 416 > $R/pmc_w.log 2>&1; echo "pmc_w rc=$?" >> $R/pmc_w.log
python $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmc_w $R/pmc_write.json > $R/pmc_write.txt 2>&1; rm -rf /tmp/pmc_w
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/probe_traffic.py 416 > $R/traffic_plain.log 2>&1
timeout -k 10 300 python bench.py --no-ladder --cpu-baseline-seconds 0 > $R/bench_short.json 2> $R/bench_short.err
