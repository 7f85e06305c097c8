#!/bin/bash
# round 4, GPU call G: parity + timing variants of the group backward (results of the EXP builds are invalid by construction)
cd /root/repo; mkdir -p gpurun_out/r04g; O=gpurun_out/r04g
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -k "group_backward" 2>&1 | tail -3
for v in "" ${VARIANTS:-cb_now cb_nomfma}; do
  if [ -z "$v" ]; then L=""; else L=build/variants/$v/libhosrender.so; fi
  echo "== variant '${v:-default}'"
  env ${L:+HOS_LIB_PATH=$L} timeout 300 python scripts/bench_chainbwd.py 262144 20 2>&1 | grep "chain_bwd=" | tail -2 | tee -a $O/variants.txt
done
