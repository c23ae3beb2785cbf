set -u
mkdir -p gpurun_out/c7
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c7/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c7/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles.sh mixed > gpurun_out/c7/collect.log 2>&1; tail -30 gpurun_out/c7/collect.log | cut -c1-200
timeout 100 python tools/prof_steps.py 2 1 mixed > gpurun_out/profiles/steps_mixed_b2.txt 2>&1
timeout 100 python tools/prof_steps.py 2 1 fp16 > gpurun_out/profiles/steps_fp16_b2.txt 2>&1
timeout 100 python tools/prof_steps.py 1 3 mixed > gpurun_out/profiles/steps_mixed_b1_n3.txt 2>&1
