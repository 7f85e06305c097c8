python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -3
