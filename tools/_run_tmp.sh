timeout 60 python tools/prof_steps.py 2 -v 2>&1 | grep -E "ms per batch|step pack|conv1_1|conv1_2" | head -5
RTP_CONV1_DIRECT=0 timeout 60 python tools/prof_steps.py 2 -v 2>&1 | grep -E "ms per batch|step pack|conv1_1 " | head -4
timeout 300 python -m pytest tests -x -q -m gpu -k "conv_stack or frame_pipeline or determinism or end_to_end or config1 or smoke or weights" 2>&1 | tail -3
