#!/bin/bash
# round 4, GPU call K: two-group variant of the group backward: parity + microbench A/B
cd /root/repo; mkdir -p gpurun_out/r04k; O=gpurun_out/r04k
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -k "group_backward" 2>&1 | tail -3
for g in 3 2; do echo "== groups $g"; HOS_CHAIN_BWD_GROUPS=$g timeout 300 python scripts/bench_chainbwd.py 262144 20 2>&1 | grep "chain_bwd=1" | tail -1 | tee -a $O/groups.txt; done
HOS_CHAIN_BWD_GROUPS=2 timeout 300 python scripts/bench_chainbwd.py 524288 20 2>&1 | grep "chain_bwd=" | tail -2 | tee -a $O/groups.txt
