#!/bin/bash
# round 6, call 28: the attention suite with the rank-one safety test
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c28; mkdir -p $O
timeout 1200 python -m pytest tests/test_autoint_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt | cut -c1-200
