#!/bin/bash
# round 5, GPU call Q: WGRAD alone, r04 tree vs HEAD (is the residual 2 % the kernel or the state the preceding launches leave?);
# TCC counter names of this box; bench line after the WGRAD address fix
cd /root/repo; mkdir -p gpurun_out/r05q; O=gpurun_out/r05q
run() { GM=131072 GONLY="$1" timeout 300 python scripts/bench_gemmp.py 40 2>&1 | grep planes; }
for rep in 1 2 3; do
  echo "== r04 tree wgrad alone"; (cd build/r04tree && run wgrad)
  echo "== HEAD wgrad alone"; run wgrad
done | tee $O/wgrad_alone.txt
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/counters.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05q/bench.json') if l.startswith('{')][-1])
print('stage3 ms', d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'])
print('stage2 ms', d['stages']['stage2']['ms_per_step'], 'stage1 ms', d['stages']['stage1']['ms_per_step'], 'infer rays/s', d['stages']['infer_1080p']['value'])
for k in d['kernels'][:8]: print('   ', k['kernel'], k['launches'], round(k['avg_us'],1))
PY
