#!/bin/bash
# round 5, call 15: eight-deep member loop for hot rows in the finishing launch — suites + Zipf / uniform kernel tables
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_fused_gpu.py tests/test_headline_gpu.py -m gpu -x -q > gpurun_out/c15_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/c15_tests.txt | cut -c1-300
bash tools_prof.sh c15_zipf --dist zipf --steps 100 --warmup 10 --no-parity | head -5
bash tools_prof.sh c15_uniform --steps 100 --warmup 10 --no-parity | head -5
