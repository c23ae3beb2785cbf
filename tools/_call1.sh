set -u
mkdir -p gpurun_out/c1
O=gpurun_out/c1
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 400 python bench.py > $O/bench_mixed.json 2> $O/bench_mixed.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/c1/bench_mixed.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('by_mfma_passes'), {k:v.get('value') for k,v in d['sub_results'].items() if isinstance(v,dict)})
P
timeout 120 python tools/prof_steps.py 2 1 mixed > $O/steps_mixed_b2.txt 2>&1
timeout 120 python tools/prof_steps.py 4 1 mixed > $O/steps_mixed_b4.txt 2>&1
timeout 120 python tools/prof_steps.py 1 1 mixed > $O/steps_mixed_b1.txt 2>&1
head -30 $O/steps_mixed_b2.txt
head -8 $O/steps_mixed_b4.txt
for bf in "4 8" "4 12" "1 8" "2 12"; do set -- $bf; timeout 120 python bench.py --no_cpu_baseline --no_sub_results --min_seconds 1.0 --batch_frames $1 --in_flight $2 > $O/bench_b$1_f$2.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('$O/bench_b$1_f$2.json') if l.startswith('{')][-1]);print('B',$1,'inflight',$2,d['value'],d['latency_ms'])"; done
