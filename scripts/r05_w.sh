#!/bin/bash
# round 5, GPU call W: full GPU suite on the final planes GEMM sources; PMC passes (GEMM traffic at the step's shapes, whole-step traffic of stages 2 / 3)
cd /root/repo; mkdir -p gpurun_out/r05w; O=gpurun_out/r05w
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -40 | tee $O/pytest.txt
bash scripts/pmc_gemmp_step.sh > $O/pmc_gemmp.log 2>&1
bash scripts/pmc_step_traffic.sh stage2 > $O/pmc_stage2.log 2>&1
bash scripts/pmc_step_traffic.sh stage3 > $O/pmc_stage3.log 2>&1
ls gpurun_out/pmc_gemmp/*.json gpurun_out/pmc_step_stage2/traffic.json gpurun_out/pmc_step_stage3/traffic.json
