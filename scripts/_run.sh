python -m pytest tests -x -q -m gpu -k "human or stage2 or deconv or split_backward" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_s2i -- python /root/repo/bench.py --primary stage2 --only-primary --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_s2i.log 2>&1
cd /root/repo; grep "dpre\|gemm_kernel<128, 128, 0>\|gemm_kernel<32, 128, 0>" gpurun_out/prof_s2i/*/*kernel_stats.csv | cut -c1-200
