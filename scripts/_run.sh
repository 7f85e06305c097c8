GM=4096 python scripts/bench_gemmp.py 3 2>&1 | grep "relu bits\|diff\|err"
bash scripts/pmc_gemmp_r02.sh > gpurun_out/pmc_r02.log 2>&1; tail -50 gpurun_out/pmc_r02.log
