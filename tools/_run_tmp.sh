PROBE_ITERS=1500 python tools/race_probe_post.py load 2>&1 | tail -3
python tools/_dbg_fused.py host 2>&1 | tail -2
python tools/_dbg_fused.py frame 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
