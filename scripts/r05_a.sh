#!/bin/bash
# round 5, GPU call A: persistent planes GEMM (HOS_GEMMP_PERSIST=1, default) against the one-workgroup-per-tile launch (=0), same box,
# interleaved rounds; accuracy lines of scripts/bench_gemmp.py; then the background parity tests on the new kernel.
cd /root/repo; mkdir -p gpurun_out/r05a; O=gpurun_out/r05a
for rep in 1 2 3; do
for p in 1 0; do
  echo "== persist=$p (round $rep)"
  HOS_GEMMP_PERSIST=$p GM=131072 GONLY="fwd(2fmt),fwd(f16),dgrad(bits),wgrad" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes
done; done | tee $O/persist_ab.txt
echo "== accuracy persist=1" | tee -a $O/persist_ab.txt
HOS_GEMMP_PERSIST=1 GM=131072 timeout 300 python scripts/bench_gemmp.py 5 2>&1 | tail -12 | tee -a $O/persist_ab.txt
echo "== small shapes" | tee -a $O/persist_ab.txt
for p in 1 0; do
  HOS_GEMMP_PERSIST=$p GM=262144 GN=256 GK=256 GONLY="fwd(2fmt),fwd(f16),dgrad(bits),wgrad" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes | sed "s/^/p=$p /"
  HOS_GEMMP_PERSIST=$p GM=131072 GN=1024 GK=576 GONLY="fwd(2fmt),fwd(f16)" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes | sed "s/^/p=$p /"
done | tee -a $O/persist_ab.txt
timeout 900 python -m pytest tests/test_gpu_round2_kernels.py tests/test_gpu_bkgd.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee $O/bench.txt
