bash scripts/pmc_gemmp_r02.sh > gpurun_out/pmc_r02.log 2>&1; tail -12 gpurun_out/pmc_r02.log
python bench.py --only-primary --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'])"
