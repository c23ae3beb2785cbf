set -u
bash tools/collect_profiles.sh mixed nopmc > gpurun_out/collect8.log 2>&1; tail -5 gpurun_out/collect8.log | cut -c1-300
python - <<'P'
import json
for f in ('bench_mixed_n1.json','bench_mixed_n1_under_rocprof.json'):
    d=json.loads([l for l in open('gpurun_out/profiles/'+f) if l.startswith('{')][-1])
    r=d['roofline']
    print(f, round(d['value'],1), 'frac', round(r['frac'],4), 'ms', round(r['ms_per_launch'],5), {k:round(v['ms_per_launch'],5) for k,v in r['by_mfma_passes'].items()}, 'pipelined', r.get('pipelined_span',{}).get('by_mfma_passes'), 'solo', round(r['solo']['ms_per_launch'],5))
P
head -5 gpurun_out/profiles/bench_mixed_n1_kernel_stats.csv | cut -c1-200
timeout 100 python bench.py --num_scales 3 --scale_gap 0.15 --no_cpu_baseline --no_sub_results > gpurun_out/profiles/bench_mixed_3scales.json 2>/dev/null; tail -c 600 gpurun_out/profiles/bench_mixed_3scales.json | cut -c1-300
timeout 100 python bench.py --model mpi --no_cpu_baseline --no_sub_results > gpurun_out/profiles/bench_mixed_mpi.json 2>/dev/null; python -c "
import json
for f in ('bench_mixed_3scales.json','bench_mixed_mpi.json'):
    d=json.loads([l for l in open('gpurun_out/profiles/'+f) if l.startswith('{')][-1]); print(f, round(d['value'],1), d['latency_ms'], d['roofline']['frac'])"
