cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_512 -- python /root/repo/bench.py --only-primary --rays 512 --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_512.log 2>&1
cd /root/repo; grep '^{' gpurun_out/prof_512.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('512 under rocprof', d['ms_per_step'])"
