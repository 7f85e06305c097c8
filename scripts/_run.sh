python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/bench_r02_final3.json 2> gpurun_out/bench_r02_final3.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r02_final3.json') if l.startswith('{')][-1])
print('stage3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel'])
for k,v in d.get('stages',{}).items(): print(k, v['value'], v['ms_per_step'], v.get('speedup_vs_torch_rocm'))
print(d['speedup_vs_torch_rocm'], d['cpu_baseline']['value'])
PY
cd /tmp && export TMPDIR=/tmp
for st in stage3 stage2 stage1; do rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_fin_$st -- python /root/repo/bench.py --primary $st --only-primary --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_fin_$st.log 2>&1; done
