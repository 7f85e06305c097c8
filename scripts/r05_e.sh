#!/bin/bash
# round 5, GPU call E: full GPU suite (no -x) after moving the guard MAX-reduce out of the captured optimiser step
cd /root/repo; mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -60 | tee $O/pytest.txt
