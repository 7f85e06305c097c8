#!/bin/bash
# round 5, GPU call AF: 40 stage-3 training steps as TWO ranks on the one GPU (gloo transport), volume decoder replicated vs sharded: the loss
# trajectories must agree (same seeds, same items); timing is not meaningful (two processes share the GPU, the sharded form runs eagerly over gloo)
cd /root/repo; mkdir -p gpurun_out/r05af; O=gpurun_out/r05af
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 HOS_BENCH_ONE_GPU=1
for sh in 0 1; do
  HOS_SHARD_DECODER=$sh timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2955$sh bench.py --gpus 2 --steps 40 --warmup 2 --only-primary --no-kernel-events 2>$O/err_$sh.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('HOS_SHARD_DECODER=$sh', 'n_gpus', d['n_gpus'], 'ms_per_step', round(d['ms_per_step'],2), 'final_loss', d['final_loss'], '|', d['launch'][:90])"
done | tee $O/two_ranks.txt
tail -3 $O/err_1.txt
