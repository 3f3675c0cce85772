cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04hist; R=$GRAFT_REPO_ROOT/gpurun_out/r04hist
timeout -k 10 300 python tools/probe_hist_cost.py 104 416 > $R/hist_cost.log 2>&1; echo "rc=$?" >> $R/hist_cost.log; cat $R/hist_cost.log
