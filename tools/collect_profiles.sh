#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root: bench line, rocprofv3 kernel stats of the
# same command, PMC passes of the dominant kernel.  Output under gpurun_out/profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py > $O/bench_fp16_n1.json 2> $O/bench_fp16_n1.err
tail -c 600 $O/bench_fp16_n1.json
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no_cpu_baseline > $O/bench_fp16_n1_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_fp16_n1_kernel_stats.csv
python $R/tools/timeline.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) > $O/bench_fp16_n1_timeline.txt 2>&1
B=$(python -c "import json;print(json.load(open('$O/bench_fp16_n1.json'))['config'].get('batch_frames', 2))" 2>/dev/null || echo 2)
: > $O/dominant_conv_pmc.txt
echo "# rocprofv3 --kernel-trace --pmc <group> -- python tools/prof_dominant.py fp16 20 $B   (one group per pass)" >> $O/dominant_conv_pmc.txt
for grp in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do  # ONE counter per pass (two TCC counters in a pass hung)
  rm -rf /tmp/pmc; timeout 40 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc -- python $R/tools/prof_dominant.py fp16 20 $B > /tmp/pmc.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc conv_ring >> $O/dominant_conv_pmc.txt 2>&1 || echo "group '$grp' failed" >> $O/dominant_conv_pmc.txt
done
tail -3 /tmp/pmc.log >> $O/dominant_conv_pmc.txt
cat $O/dominant_conv_pmc.txt
