#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root: bench line, rocprofv3 kernel stats of the
# same command, PMC passes of the dominant kernel.  Output under gpurun_out/profiles/ (copy into profiles/ as rNN_*).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles
P=${1:-mixed}          # precision mode of the headline line (bench.py default)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --precision $P > $O/bench_${P}_n1.json 2> $O/bench_${P}_n1.err
cp $R/bench_detail.json $O/bench_detail_${P}_n1.json 2>/dev/null
echo "bench line: $(wc -c < $O/bench_${P}_n1.json) bytes"; tail -c 400 $O/bench_${P}_n1.json
( cd $R && timeout 120 python tools/stamp_timeline.py host 2 7 800 2>&1 | grep -v amdgpu.ids ) > $O/stamp_timeline.txt
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --precision $P --no_cpu_baseline --no_sub_results --no_parity > $O/bench_${P}_n1_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_${P}_n1_kernel_stats.csv
python $R/tools/timeline.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) > $O/bench_${P}_n1_timeline.txt 2>&1
[ "${2:-}" = nopmc ] && exit 0   # bench line + kernel stats only
B=$(python -c "import json;print(json.loads([l for l in open('$O/bench_${P}_n1.json') if l.startswith('{')][-1])['config'].get('batch_frames', 2))" 2>/dev/null || echo 2)
: > $O/dominant_conv_pmc_${P}_b$B.txt
echo "# rocprofv3 --kernel-trace --pmc <group> -- python tools/prof_dominant.py $P 20 $B   (one group per pass)" >> $O/dominant_conv_pmc_${P}_b$B.txt
for grp in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do  # ONE counter per pass (two TCC counters in a pass hung)
  rm -rf /tmp/pmc; timeout 60 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc -- python $R/tools/prof_dominant.py $P 20 $B > /tmp/pmc.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc conv_ring >> $O/dominant_conv_pmc_${P}_b$B.txt 2>&1 || echo "group '$grp' failed" >> $O/dominant_conv_pmc_${P}_b$B.txt
done
tail -3 /tmp/pmc.log >> $O/dominant_conv_pmc_${P}_b$B.txt
cat $O/dominant_conv_pmc_${P}_b$B.txt | cut -c1-150
if [ "$P" = mixed ]; then  # the same shape as an fp8-compensated launch (stages 4-6: most of the symbol's time in the mixed plan)
  Q=$O/dominant_conv_pmc_${P}_b${B}_2q.txt
  echo "# RTP_DOMINANT_Q=1 rocprofv3 --kernel-trace --pmc <group> -- python tools/prof_dominant.py $P 20 $B   (one group per pass; the fp8-compensated launch)" > $Q
  for grp in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
    rm -rf /tmp/pmc; RTP_DOMINANT_Q=1 timeout 60 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc -- python $R/tools/prof_dominant.py $P 20 $B > /tmp/pmc.log 2>&1
    python $R/tools/pmc_summary.py /tmp/pmc conv_ring >> $Q 2>&1 || echo "group '$grp' failed" >> $Q
  done
  tail -1 /tmp/pmc.log >> $Q
  cat $Q | cut -c1-150
fi
