#!/bin/bash
# round 4, call 18: CIN unit tests (the large cases with near-kink units masked) + the CIN kernels' whole test set
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c18
O=gpurun_out/r4c18
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_bf16_gpu.py tests/test_kernels_gpu.py tests/test_headline_gpu.py -m gpu -q -k "cin or xdeepfm or bf16 or xDeepFM" > $O/pytest.log 2>&1
grep -E "passed|failed|error|FAILED" $O/pytest.log | tail -8
