cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $R/tools/prof_frame.py fp16 1 > /tmp/pf.log 2>&1; grep "conv" /tmp/pf.log | tail -1
f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
python $R/tools/stats_nonconv.py $f
