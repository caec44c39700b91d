#!/bin/bash
# round 3, GPU call 29: the N > 1 steps' RCCL calls really issued on one rank (force_collectives)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_parallel_gpu.py -q -m gpu -k "sharded or parallel or rccl" 2>&1 | tail -25 | cut -c1-220
