#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c12
O=gpurun_out/r4c12
for i in 1 2 3; do timeout 300 python -m pytest tests/test_compiled_gpu.py -q -k "DCN-binary-False or DeepFM-binary-False" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head -12; done > $O/t_flaky.txt
cat $O/t_flaky.txt
