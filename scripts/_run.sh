python -m pytest tests/test_gpu_human.py -x -q 2>&1 | tail -2
for v in 1 0 1 0; do echo "DPE_THIN=$v $(HOS_DPE_THIN=$v python bench.py --primary stage2 --only-primary --steps 100 --warmup 10 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_dcat2 -- python /root/repo/bench.py --primary stage2 --only-primary --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_dcat2.log 2>&1
