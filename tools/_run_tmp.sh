RTP_RING_SPEC=1 timeout 60 python tools/prof_steps.py 1 2>&1 | head -6
RTP_RING_SPEC=1 timeout 60 python tools/prof_steps.py 2 2>&1 | head -5
RTP_RING_SPEC=1 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_stack or determinism or frame_pipeline or batching" 2>&1 | tail -4
