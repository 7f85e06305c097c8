#!/bin/bash
# round 5, GPU call D: full GPU suite on {persistent planes GEMM, bf16-only NeRF-MLP forward, self-re-arming range guard}; bench A/B of the
# forward format (HOS_NERF_BF16_FWD=1 default / 0 = fp16 planes + second bf16 epilogue)
cd /root/repo; mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee $O/pytest.txt
for f in 1 0 1 0; do
  echo "== HOS_NERF_BF16_FWD=$f"
  HOS_NERF_BF16_FWD=$f timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('stage3 ms', d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'])
print('stage2 ms', d['stages']['stage2']['ms_per_step'], 'stage1 ms', d['stages']['stage1']['ms_per_step'], 'infer rays/s', d['stages']['infer_1080p']['value'])
for k in d['kernels'][:8]: print('   ', k['kernel'], k['launches'], round(k['avg_us'],1))
"
done | tee $O/bench_ab.txt
