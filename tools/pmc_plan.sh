#!/bin/bash
# HBM traffic of the dominant launch of ONE plan: rocprofv3 PMC passes (one counter per pass, --kernel-trace only) over tools/prof_dominant.py.
#   tools/pmc_plan.sh PREC BATCH SCALES MODEL OUTFILE      e.g.  tools/pmc_plan.sh mixed 1 3 coco gpurun_out/r05_dominant_conv_pmc_mixed_b1_n3.txt
# Writes OUTFILE (plain launch) and ${OUTFILE%.txt}_2q.txt (the fp8-compensated launch of the same shape); bench.py's pmc_traffic() picks the files up
# from profiles/ by their header line.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
P=$1; B=$2; N=$3; M=$4; OUT=$5
cd /tmp && export TMPDIR=/tmp
for q in "" 1; do
  F=$R/$OUT; [ -n "$q" ] && F=$R/${OUT%.txt}_2q.txt
  echo "# ${q:+RTP_DOMINANT_Q=1 }rocprofv3 --kernel-trace --pmc <group> -- python tools/prof_dominant.py $P 20 $B $N $M   (one group per pass${q:+; the fp8-compensated launch})" > $F
  for grp in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc
    RTP_DOMINANT_Q=$q timeout 90 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc -- python $R/tools/prof_dominant.py $P 20 $B $N $M > /tmp/pmc.log 2>&1
    python $R/tools/pmc_summary.py /tmp/pmc conv_ring >> $F 2>&1 || echo "group '$grp' failed" >> $F
  done
  tail -1 /tmp/pmc.log >> $F
done
