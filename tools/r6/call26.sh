#!/bin/bash
# round 6, call 26: the AutoInt graph through the captured loop (new test) + the compiled suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c26; mkdir -p $O
timeout 1200 python -m pytest tests/test_compiled_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
