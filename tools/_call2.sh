set -u
O=gpurun_out/c2; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_probe tools/dpp_probe.hip 2>/dev/null && /tmp/dpp_probe | tee $O/dpp_probe.txt
timeout 300 python tools/ab_hash.py > $O/ab_base.txt 2>&1
RTP_RING_ILV=1 timeout 300 python tools/ab_hash.py > $O/ab_ilv.txt 2>&1
cat $O/ab_base.txt; echo; cat $O/ab_ilv.txt
if diff <(grep -v '^#' $O/ab_base.txt) <(grep -v '^#' $O/ab_ilv.txt) > /dev/null; then echo "ILV BIT-IDENTICAL"; else echo "ILV DIFFERS"; fi
for v in 0 1; do
  RTP_RING_ILV=$v timeout 120 python tools/prof_steps.py 2 1 mixed > $O/steps_mixed_b2_ilv$v.txt 2>&1
  RTP_RING_ILV=$v timeout 120 python tools/prof_steps.py 2 1 fp16 > $O/steps_fp16_b2_ilv$v.txt 2>&1
  head -6 $O/steps_mixed_b2_ilv$v.txt; head -4 $O/steps_fp16_b2_ilv$v.txt
done
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --no_cpu_baseline --no_sub_results --min_seconds 1.5 > $O/bench_$name.json 2>/dev/null
  python -c "
import json;d=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]);print('$name',round(d['value'],1),'p50',round(d['latency_ms']['p50_pipelined'],2),'dom ms',d['roofline'].get('by_mfma_passes'))" 2>&1 | cut -c1-400
}
run base RTP_X=0
run ilv RTP_RING_ILV=1
run base2 RTP_X=0
run ilv2 RTP_RING_ILV=1
run ilv_postlow RTP_RING_ILV=1 RTP_POST_PRIO=1
run ilv_convhigh RTP_RING_ILV=1 RTP_CONV_PRIO=-1
run ilv_skippost RTP_RING_ILV=1 RTP_DIAG_SKIP_POST=2
