bash scripts/pmc_gemmp_r02.sh 2>&1 | tail -70
