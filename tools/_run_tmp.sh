timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
PROBE_ITERS=600 python tools/race_probe_post.py load 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $R/tools/prof_frame.py fp16 1 > /tmp/pf.log 2>&1; grep "conv" /tmp/pf.log | tail -1
python $R/tools/stats_nonconv.py $(find /tmp/pf -name "*kernel_stats.csv" | head -1)
cd $R
P='import sys,json
d=json.loads(sys.stdin.read()); print(round(d["value"],1), "fps  p50", round(d["latency_ms"]["p50_pipelined"],2), "ms  roof", round(d["roofline"]["frac"],3), "whole", round(d["conv_stack_whole_frame"]["frac"],3), d["stage_ms_last_frame"])'
run() { echo "== B=$B F=$F N=${N:-1} $*"; env "$@" timeout 300 python bench.py --no_cpu_baseline --steps 600 --warmup 60 --batch_frames $B --in_flight $F --num_scales ${N:-1} --scale_gap 0.15 2>&1 | tail -1 | python -c "$P"; }
B=1 F=8;  run X=1
B=2 F=8; run X=1
