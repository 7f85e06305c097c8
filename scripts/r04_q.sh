#!/bin/bash
# round 4, GPU call Q: the whole GPU suite + smoke() of the current tree
cd /root/repo; mkdir -p gpurun_out/r04q; O=gpurun_out/r04q
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed|error" $O/gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
