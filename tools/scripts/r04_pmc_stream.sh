# HBM traffic and SQ counters of the command-stream kernel (free-running year), separate --pmc passes; kernel stats of the phosphorus years
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04pmcs; R=$GRAFT_REPO_ROOT/gpurun_out/r04pmcs
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/probe_traffic_stream.py 416 > $R/plain.log 2>&1; echo "rc=$?" >> $R/plain.log; tail -2 $R/plain.log | cut -c1-600
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  tag=$(echo $pass | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmcs_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_traffic_stream.py 416 > $R/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?" >> $R/pmc_$tag.log
  python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmcs_$tag $R/pmc_$tag.json > $R/pmc_$tag.txt 2>&1; rm -rf /tmp/pmcs_$tag
  tail -1 $R/pmc_$tag.log; grep -h "years_as_command_streams" $R/pmc_$tag.log | cut -c1-300
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ph -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_stream_phos.py 416 > $R/phos_rocprof.log 2>&1; echo "rc=$?" >> $R/phos_rocprof.log
find /tmp/prof_ph -name "*kernel_stats.csv" -exec cp {} $R/phos_kernel_stats.csv \;
rm -rf /tmp/prof_ph
head -6 $R/phos_kernel_stats.csv | cut -c1-150
