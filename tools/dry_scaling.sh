#!/bin/bash
# Host side of rtpose.bin --num_gpu N at N x the measured per-GPU rates, no GPU work.
# usage: tools/dry_scaling.sh [rate_1scale=984] [rate_3scales=372]   -> stdout (commit under profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}
R1=${1:-984}; R3=${2:-372}
T=$(mktemp -d)
echo "# host: $(nproc) hardware threads; rtpose.bin --video synthetic:1280x720:<frames> --model coco --no_frame_drops --no_display --frames_in_flight 8 --write_json DIR --dry_engine RATE --num_gpu N"
echo "# columns: N, per-worker rate, people per frame, json_writers, producer_threads -> frames/s first frame committed -> last frame written | target N*rate | per-worker frames"
run() {  # N rate people jw pt frames
  rm -rf $T/j; mkdir -p $T/j
  out=$($R/caffe_rtpose_amd/rtpose.bin --video synthetic:1280x720:$6 --model coco --num_gpu $1 --dry_engine $2 --dry_people $3 --json_writers $4 --producer_threads $5 \
        --write_json $T/j --no_frame_drops --no_display --frames_in_flight 8 2>&1)
  fps=$(echo "$out" | grep -o "[0-9.]* FPS first frame" | cut -d' ' -f1)
  pw=$(echo "$out" | grep -o "processed [0-9]* frames" | cut -d' ' -f2 | tr '\n' ' ')
  echo "N=$1 rate=$2 people=$3 json_writers=$4 producer_threads=$5 -> $fps frames/s | target $(( $1 * ${2%.*} )) | $pw| files $(ls $T/j | wc -l)"
}
for n in 1 2 4 8; do run $n $R1 5 -1 0 $(( 3000 * n )); done
run 8 $R1 76 -1 0 24000          # noise-map worst case: 76 people per frame
run 8 $R1 76 0 0 24000           # the reference's structure: ONE display thread writes every file
run 8 $R1 76 -1 2 12000          # two producer threads only
run 8 $R3 20 -1 0 9000           # 3 scales
run 16 $R1 5 -1 0 32000          # headroom: twice the node
rm -rf $T
