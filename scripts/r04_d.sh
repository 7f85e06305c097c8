#!/bin/bash
# round 4, GPU call D: full GPU suite of the tree + bench lines (4096 / 512 rays) + stage kernel stats
cd /root/repo; mkdir -p gpurun_out/r04d; O=gpurun_out/r04d
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
python bench.py --no-cpu-baseline --no-torch-baseline > $O/bench_4096.json 2> $O/bench_4096.err; tail -c 1500 $O/bench_4096.json
python bench.py --no-cpu-baseline --no-torch-baseline --no-infer --rays 512 > $O/bench_512.json 2> $O/bench_512.err; tail -c 600 $O/bench_512.json
