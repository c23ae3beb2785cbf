timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
P='import sys,json
d=json.loads(sys.stdin.read()); print(round(d["value"],1), "fps  p50", round(d["latency_ms"]["p50_pipelined"],2), "ms  roof", round(d["roofline"]["frac"],3), "whole", round(d["conv_stack_whole_frame"]["frac"],3), d["stage_ms_last_frame"])'
run() { echo "== B=$B F=$F $*"; env "$@" timeout 300 python bench.py --no_cpu_baseline --steps 600 --warmup 60 --batch_frames $B --in_flight $F 2>&1 | tail -1 | python -c "$P"; }
B=1 F=8;  run X=1; run RTP_DIAG_SKIP_POST=1
B=1 F=4;  run X=1
B=2 F=8; run X=1
B=4 F=16; run X=1; run RTP_DIAG_SKIP_POST=1
B=4 F=8; run X=1
B=6 F=18; run X=1
