python -m pytest tests/test_gpu_human.py tests/test_gpu_dist.py tests/test_gpu_speedup.py tests/test_gpu_soak.py tests/test_gpu_split_backward.py -x -q 2>&1 | tail -3
for i in 1 2; do
echo "new  $(python bench.py --primary stage2 --only-primary --steps 100 --warmup 10 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
echo "prev $(cd build/prev_tree && python bench.py --primary stage2 --only-primary --steps 100 --warmup 10 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
