set -u
O=gpurun_out/c3; mkdir -p $O
# parity of the new geometry: the conv-stack / taps / pipeline tests + precision
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_precision.py tests/test_ref_golden.py -x -q > $O/pytest_geom.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_geom.txt
RTP_HALO_SHARED=0 timeout 300 python tools/ab_hash.py > $O/ab_old.txt 2>&1
timeout 300 python tools/ab_hash.py > $O/ab_new.txt 2>&1
RTP_RING_ILV=1 timeout 300 python tools/ab_hash.py > $O/ab_new_ilv.txt 2>&1
cat $O/ab_old.txt $O/ab_new.txt $O/ab_new_ilv.txt
timeout 120 python tools/prof_steps.py 2 1 mixed > $O/steps_mixed_b2.txt 2>&1; head -8 $O/steps_mixed_b2.txt
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --no_cpu_baseline --no_sub_results --min_seconds 1.5 > $O/bench_$name.json 2>/dev/null
  python -c "
import json;d=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]);print('$name',round(d['value'],1),'p50',round(d['latency_ms']['p50_pipelined'],2),'p95',round(d['latency_ms']['p95_pipelined'],2))" 2>&1 | cut -c1-400
}
run old RTP_HALO_SHARED=0
run new RTP_X=0
run new_own0 RTP_POST_OWN0=1
run new_cus8 RTP_POST_CUS=8
run new_cus8_own0 RTP_POST_CUS=8 RTP_POST_OWN0=1
run new_cus16_own0 RTP_POST_CUS=16 RTP_POST_OWN0=1
run new_skippost RTP_DIAG_SKIP_POST=2
run new2 RTP_X=0
