#!/bin/bash
# Everything profiles/<round>_* of a tree is made from, in one GPU call:   scripts/collect.sh r06 [outdir]
#   smoke, the default bench line, the strong-scaling ray sweep (4096 / N rays = one rank's share at N = 1, 2, 4, 8), the one-GPU
#   timing model of rank 0 of 8 with the sharded decoder, and rocprofv3 --kernel-trace --stats of the replayed steps of every stage.
# Outputs land under gpurun_out/<round>final/ and gpurun_out/<round>_*_kernel_stats.csv; copy what is to be judged into profiles/.
RND=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=${2:-gpurun_out/${RND}final}; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_${RND}_final.json 2> $O/bench.err
for r in 512 1024 2048 4096; do
  timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1
done > $O/${RND}_strong_scaling_sweep.jsonl
for r in 512 4096; do HOS_MODEL_SHARD=8 timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1; done > $O/${RND}_model_shard8.jsonl
HOS_TWO_STREAMS=0 bash scripts/prof_step.sh 4096 ${RND}_stage3_one_stream
bash scripts/prof_step.sh 4096 ${RND}_stage3
bash scripts/prof_step.sh 512 ${RND}_stage3_512rays
PRIMARY=stage2 bash scripts/prof_step.sh 2048 ${RND}_stage2
PRIMARY=stage1 bash scripts/prof_step.sh 1024 ${RND}_stage1
RND=$RND O=$O python - <<'PY'
import json, os
rnd, o = os.environ["RND"], os.environ["O"]
d = json.loads([l for l in open(f"{o}/bench_{rnd}_final.json") if l.startswith("{")][-1])
print("stage3 ms", d["ms_per_step"], d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["avg_us"], d["roofline"]["frac"], d["roofline"]["traffic"], "x torch", d.get("speedup_vs_torch_rocm"))
for k, v in d["stages"].items():
    print(k, v.get("ms_per_step"), v.get("value"), v.get("speedup_vs_torch_rocm"), v.get("error"))
for k in d["kernels"][:6]:
    print("   ", k["kernel"], k["launches"], round(k["avg_us"], 1), round(k["tflops"], 1))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["thread_sweep_rays_per_s"], d["cpu_baseline"]["step_seconds"])
for f in (f"{rnd}_strong_scaling_sweep.jsonl", f"{rnd}_model_shard8.jsonl"):
    for l in open(f"{o}/{f}"):
        x = json.loads(l); print(f, x["config"]["global_rays"], round(x["ms_per_step"], 3))
PY
