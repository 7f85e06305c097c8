for sd in 1 2 3 4 5 6; do SOAK_GEMM=fp32 python scripts/soak_poison.py 2 500 $sd 2048 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300; done
for sd in 61 62; do python scripts/soak_poison.py 3 500 $sd 4096 2>&1 | grep -v amdgpu | tail -1; done
for sd in 71 72; do python scripts/soak_poison.py 1 1500 $sd 1024 2>&1 | grep -v amdgpu | tail -1; done
