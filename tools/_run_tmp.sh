timeout 60 python tools/prof_steps.py 1 2>&1 | head -3
RTP_RING_K4=1 timeout 60 python tools/prof_steps.py 1 2>&1 | head -3
RTP_RING_K4=1 timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_stack or determinism" 2>&1 | tail -3
