#!/bin/bash
# round 5, call 5: the election with the BIG template (partition bytes beyond 8192 rows) — fused / headline suites, kernel tables
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_compiled_gpu.py -m gpu -x -q > gpurun_out/c5_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/c5_tests.txt
timeout 300 python tools/r5/dbg_elect.py > gpurun_out/c5_dbg.txt 2>&1; grep -c " ok " gpurun_out/c5_dbg.txt; grep "BAD\|MISMATCH\|Error" gpurun_out/c5_dbg.txt | head
bash tools_prof.sh c5_b8192 --steps 100 --warmup 10 --no-parity | head -7
bash tools_prof.sh c5_b32768 --batch 32768 --steps 50 --warmup 10 --no-parity | head -7
bash tools_prof.sh c5_b65536 --batch 65536 --steps 30 --warmup 10 --steps-per-graph 5 --no-parity | head -7
