python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r02_final6.json 2> gpurun_out/bench_r02_final6.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r02_final6.json') if l.startswith('{')][-1])
print('stage3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel'])
for k,v in d.get('stages',{}).items(): print(k, v['value'], v['ms_per_step'], v.get('speedup_vs_torch_rocm'))
print(d['speedup_vs_torch_rocm'], d['cpu_baseline']['value'])
PY
cd /tmp && export TMPDIR=/tmp
for st in stage3 stage2 stage1; do rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_fin6_$st -- python /root/repo/bench.py --primary $st --only-primary --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_fin6_$st.log 2>&1; done
