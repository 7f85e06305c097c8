python -m pytest tests/test_gpu_configs.py -x -q -k launcher 2>&1 | tail -4
