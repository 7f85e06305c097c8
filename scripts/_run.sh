python -m pytest tests/test_gpu_speedup.py -x -q 2>&1 | grep -E "^E  |passed|failed" | cut -c1-600 | head -12
