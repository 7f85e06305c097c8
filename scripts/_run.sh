python bench.py > gpurun_out/bench_r02_mid.json 2> gpurun_out/bench_r02_mid.err; tail -c 600 gpurun_out/bench_r02_mid.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r02_mid.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k,v in d.get('stages',{}).items(): print(k, {kk:v[kk] for kk in v if kk in ('value','ms_per_step','torch_rocm','speedup_vs_torch_rocm','rays')})
print(d.get('cpu_baseline')); print(d.get('torch_rocm_baseline') or d.get('torch_baseline'))
PY
