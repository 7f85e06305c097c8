python -m pytest tests/test_gpu_prologue.py tests/test_gpu_compact.py tests/test_gpu_stage2.py tests/test_gpu_human.py tests/test_gpu_stage3.py tests/test_gpu_speedup.py -x -q 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_try2.json 2> gpurun_out/bench_try2.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_try2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_try2.json'))
print({k:d[k] for k in ('value','ms_per_step','launch','algorithmic_tflops')}, d.get('roofline',{}).get('frac'), d.get('speedup_vs_torch_rocm'))
for k,v in d.get('stages',{}).items(): print(k, {a:v[a] for a in ('value','ms_per_step','launch','algorithmic_tflops')}, v.get('speedup_vs_torch_rocm'))
PY
