#!/bin/bash
# round 5, GPU call AH: final bench line with one workgroup per tile as the default launch; the 1080p frame with and without persistent launches
cd /root/repo; mkdir -p gpurun_out/r05final
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05final/bench_r05_final.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05final/bench_r05_final.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_us"], d["stages"]["stage2"]["ms_per_step"], d["stages"]["stage1"]["ms_per_step"], d["stages"]["infer_1080p"]["value"], d["speedup_vs_torch_rocm"], d["stages"]["stage2"]["speedup_vs_torch_rocm"])
for k in d["kernels"][:3]: print(k["kernel"], k["avg_us"], k["tflops"])
print(d["stages"]["stage3_fresh_items"]["ms_per_step"], d["stages"]["stage3_with_lpips"]["ms_per_step"], d["cpu_baseline"]["value"])
PY
for rep in 1 2; do for p in 0 1; do echo -n "infer_1080p HOS_GEMMP_PERSIST=$p: "; HOS_GEMMP_PERSIST=$p timeout 600 python scripts/bench_infer.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['rays_per_s_frame']), round(d['ms_per_frame'],1))"; done; done
