python -m pytest tests/test_gpu_soak.py -x -q 2>&1 | tail -3
for sd in 1 2 3 4; do python scripts/soak_poison.py 3 150 $sd 2>&1 | grep -v amdgpu | tail -1; done
