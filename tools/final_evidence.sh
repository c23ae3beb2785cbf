#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}   # round-end evidence on the GPU box: GPU tests (with their printed numbers), smoke, steps, CLI (1 and 4 engines), 2-rank bench, then tools/collect_profiles.sh
O=gpurun_out/final; mkdir -p $O
timeout 2400 python -m pytest tests -q -s -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( echo "# python tools/prof_steps.py 2 1 mixed -v   (every plan step alone on the chip, 30 repetitions; final round-6 state)"; timeout 200 python tools/prof_steps.py 2 1 mixed -v
  echo; echo "# python tools/prof_steps.py 2 1 fp16"; timeout 200 python tools/prof_steps.py 2 1 fp16
  echo; echo "# python tools/prof_steps.py 1 3 mixed"; timeout 200 python tools/prof_steps.py 1 3 mixed ) 2>&1 | grep -v amdgpu.ids > $O/steps.txt
head -3 $O/steps.txt
T=$(mktemp -d)
( echo "# caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:3000 --model coco --write_json DIR --no_frame_drops --no_display --frames_in_flight 7 --batch_frames 2"
  caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:3000 --model coco --write_json $T/j1 --no_frame_drops --no_display --frames_in_flight 7 --batch_frames 2 2>&1 | tail -6
  echo; echo "# ... --num_gpu 4 --devices 0,0,0,0 --share_weights (four engines on the one GPU: what --num_gpu N starts, as far as one device can show it)"
  caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:3000 --model coco --write_json $T/j4 --no_frame_drops --no_display --frames_in_flight 7 --batch_frames 2 --num_gpu 4 --devices 0,0,0,0 --share_weights 2>&1 | tail -12
  echo; echo "# ... --video synthetic:1280x720:1000 --num_scales 3 --scale_gap 0.15 --frames_in_flight 3 --batch_frames 1"
  caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:1000 --model coco --num_scales 3 --scale_gap 0.15 --write_json $T/j3 --no_frame_drops --no_display --frames_in_flight 3 --batch_frames 1 2>&1 | tail -4
  echo; echo "# ... --start_scale 0.8 --num_scales 2 --scale_gap 0.15 --resolution 640x360 (round 5: a start scale below 1, a display smaller than the net input)"
  caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:600 --model coco --start_scale 0.8 --num_scales 2 --scale_gap 0.15 --resolution 640x360 --write_json $T/j5 --no_frame_drops --no_display --frames_in_flight 4 --batch_frames 1 2>&1 | tail -4 ) 2>&1 | grep -v amdgpu.ids > $O/cli.txt
tail -3 $O/cli.txt; rm -rf $T
( echo "# python bench.py --gpus 2 --devices 0,0 --comm gloo --broadcast_weights --no_cpu_baseline --no_sub_results --no_parity   (two ranks on the one GPU)"
  timeout 300 python bench.py --gpus 2 --devices 0,0 --comm gloo --broadcast_weights --no_cpu_baseline --no_sub_results --no_parity 2>&1 | grep "^{" 
  echo; echo "# the same with --comm nccl: RCCL refuses two ranks on one device; does every rank fall back to gloo?"
  timeout 300 python bench.py --gpus 2 --devices 0,0 --comm nccl --no_cpu_baseline --no_sub_results --no_parity 2>&1 | grep "^{\|rror" | cut -c1-1500 | head -8; echo "rc $?"
  echo; echo "# python bench.py --calibrate 2 --no_cpu_baseline --no_sub_results --no_parity   (load-time calibration on the default synthetic weights)"
  timeout 300 python bench.py --calibrate 2 --no_cpu_baseline --no_sub_results --no_parity 2>&1 | grep "^{" ) > $O/bench_two_ranks_and_calibration.txt 2>&1
cut -c1-300 $O/bench_two_ranks_and_calibration.txt | head -12
bash tools/collect_profiles.sh mixed > $O/collect.log 2>&1; tail -2 $O/collect.log
bash tools/pmc_plan.sh mixed 1 3 coco $O/dominant_conv_pmc_mixed_b1_n3.txt > /dev/null 2>&1
bash tools/pmc_plan.sh mixed 5 1 mpi $O/dominant_conv_pmc_mixed_b5_mpi.txt > /dev/null 2>&1
ls $O gpurun_out/profiles
