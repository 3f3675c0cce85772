# SQ counters of the one-launch frozen year at 416 x 416 (separate --pmc passes; the process dumps core in the HSA tear-down
# after the tables are written, see profiles/README.md)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rec5; R=$GRAFT_REPO_ROOT/gpurun_out/rec5
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  tag=$(echo $pass | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_traffic.py 416 > $R/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?" >> $R/pmc_$tag.log
  python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmc_$tag $R/pmc_$tag.json > $R/pmc_$tag.txt 2>&1; rm -rf /tmp/pmc_$tag
done
ls -la $R
