#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}   # round-end evidence on the GPU box: GPU tests, precision output, smoke, bench lines (3 scales, MPI), steps, CLI, then tools/collect_profiles.sh
O=gpurun_out/final; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python -m pytest tests/test_precision.py -q -s -m gpu > $O/precision_tests.txt 2>&1; tail -2 $O/precision_tests.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --num_scales 3 --scale_gap 0.15 --no_sub_results > $O/bench_mixed_3scales.json 2>/dev/null; tail -c 300 $O/bench_mixed_3scales.json; echo
timeout 400 python bench.py --model mpi --no_sub_results > $O/bench_mixed_mpi.json 2>/dev/null; head -c 300 $O/bench_mixed_mpi.json; echo
( echo "# python tools/prof_steps.py 2 1 mixed -v   (every plan step alone on the chip, 30 repetitions; final round-3 state)"; timeout 200 python tools/prof_steps.py 2 1 mixed -v
  echo; echo "# python tools/prof_steps.py 2 1 fp16"; timeout 200 python tools/prof_steps.py 2 1 fp16
  echo; echo "# python tools/prof_steps.py 1 3 mixed"; timeout 200 python tools/prof_steps.py 1 3 mixed ) > $O/steps.txt 2>&1
head -3 $O/steps.txt
T=$(mktemp -d)
( echo "# caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:3000 --model coco --write_json DIR --no_frame_drops --no_display --frames_in_flight 7 --batch_frames 2"
  caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:3000 --model coco --write_json $T/j1 --no_frame_drops --no_display --frames_in_flight 7 --batch_frames 2 2>&1 | tail -6
  echo; echo "# ... --video synthetic:1280x720:1000 --num_scales 3 --scale_gap 0.15 --frames_in_flight 3 --batch_frames 1"
  caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:1000 --model coco --num_scales 3 --scale_gap 0.15 --write_json $T/j3 --no_frame_drops --no_display --frames_in_flight 3 --batch_frames 1 2>&1 | tail -4 ) > $O/cli.txt 2>&1
tail -3 $O/cli.txt; rm -rf $T
bash tools/collect_profiles.sh mixed > $O/collect.log 2>&1; tail -2 $O/collect.log
