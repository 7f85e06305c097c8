for sd in 1 2 3; do python scripts/soak_poison.py 1 500 $sd 1024 2>&1 | grep -v amdgpu | tail -1; done
for sd in 1 2; do python scripts/soak_poison.py 3 200 $sd 4096 2>&1 | grep -v amdgpu | tail -1; done
for sd in 21 22 23 24 25 26; do python scripts/soak_poison.py 2 500 $sd 2048 2>&1 | grep -v amdgpu | tail -1; done
