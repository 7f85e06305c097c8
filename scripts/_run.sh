python -m pytest tests/test_gpu_stress.py -x -q 2>&1 | tail -12
