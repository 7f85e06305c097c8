bash scripts/pmc_gemmp_r02.sh > gpurun_out/pmc_r02.log 2>&1; tail -3 gpurun_out/pmc_r02.log
