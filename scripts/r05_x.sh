#!/bin/bash
# round 5, GPU call X: IPE encoder with the library's sinf / expf (HEAD), hardware v_sin / v_exp (timing only), own Cody-Waite sine (+ libm exp / own exp);
# one-GPU model of rank 0 of 8 with the sharded volume decoder at 512 rays
cd /root/repo; mkdir -p gpurun_out/r05x; O=gpurun_out/r05x
for rep in 1 2; do
echo "== HEAD"; timeout 300 python scripts/bench_encode.py 2>&1 | grep encode
for v in enc_fast enc_own1 enc_own2; do echo "== $v"; HOS_LIB_PATH=build/variants/$v/libhosrender.so timeout 300 python scripts/bench_encode.py 2>&1 | grep encode; done
done | tee $O/encode.txt
for ms in 0 8; do for r in 512 4096; do
  echo -n "HOS_MODEL_SHARD=$ms rays $r: "
  HOS_MODEL_SHARD=$ms timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>&1 | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done; done | tee $O/model_shard.txt
