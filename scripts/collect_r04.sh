#!/bin/bash
# Everything profiles/r04_* is made from, in one GPU call: scripts/collect_r04.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err
for r in 512 1024 2048 4096; do
  python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1
done > gpurun_out/r04_strong_scaling_sweep.jsonl
bash scripts/prof_step.sh 4096 r04_stage3
bash scripts/prof_step.sh 512 r04_stage3_512rays
PRIMARY=stage2 bash scripts/prof_step.sh 2048 r04_stage2
bash scripts/pmc_gemmp_step.sh > gpurun_out/pmc_gemmp.log 2>&1
bash scripts/pmc_step_traffic.sh stage2 > gpurun_out/pmc_stage2.log 2>&1
bash scripts/pmc_step_traffic.sh stage3 > gpurun_out/pmc_stage3.log 2>&1
ls gpurun_out/pmc_gemmp/*.json gpurun_out/pmc_step_stage2/traffic.json gpurun_out/pmc_step_stage3/traffic.json
tail -c 300 gpurun_out/bench_r04.json
