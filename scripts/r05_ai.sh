#!/bin/bash
# round 5, GPU call AI: every r05 profile of the FINAL tree (one workgroup per tile as the default launch) re-collected: scripts/collect_r05.sh + the PMC passes
cd /root/repo
bash scripts/collect_r05.sh 2>&1 | grep -v "rocprofv3\|amdgpu.ids" | tail -22
bash scripts/pmc_gemmp_step.sh > gpurun_out/r05final/pmc_gemmp.log 2>&1
bash scripts/pmc_step_traffic.sh stage2 > gpurun_out/r05final/pmc_stage2.log 2>&1
bash scripts/pmc_step_traffic.sh stage3 > gpurun_out/r05final/pmc_stage3.log 2>&1
bash scripts/pmc_clock.sh > gpurun_out/r05final/pmc_clock.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_gemmp/pmc_gemmp_traffic.json')); print(d['source_hash'])
for k,v in d['kernels'].items(): print(k, round(v['hbm_bytes_per_launch']/1e6), round(v['hbm_bytes_per_launch']/v['algorithmic_bytes_approx'],3), v['mfma_busy_cycles']/(1024)/(v['gui_active_cycles']/8))
for st in ('stage2','stage3'): print(st, json.load(open(f'gpurun_out/pmc_step_{st}/traffic.json'))['hbm_bytes_per_step']/1e9)
c=json.load(open('gpurun_out/pmc_clock/clock.json'))
for k in ('random','zero_filled'): print(k, {t:(round(v['gui_active_cycles_per_xcd']), round(v['effective_clock_ghz'],3)) for t,v in c[k].items()})
PY
