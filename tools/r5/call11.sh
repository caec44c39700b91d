#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "chained" > gpurun_out/c11_tests.txt 2>&1
tail -40 gpurun_out/c11_tests.txt | cut -c1-300
