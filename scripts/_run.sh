bash scripts/pmc_gemmp_r02.sh > gpurun_out/pmc_gemmp.log 2>&1; tail -5 gpurun_out/pmc_gemmp.log
