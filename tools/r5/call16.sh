#!/bin/bash
# round 5, call 16: wave-aggregated list compaction in the large-batch election; the two un-spilled weight-gradient kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_kernels_gpu.py tests/test_f3_gpu.py tests/test_golden_gpu.py -m gpu -x -q > gpurun_out/c16_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/c16_tests.txt | cut -c1-300
timeout 200 python tools/r5/dbg_elect.py 2>&1 | grep -c " ok "
bash tools_prof.sh c16_b32768 --batch 32768 --steps 50 --warmup 10 --no-parity | head -6
bash tools_prof.sh c16_b65536 --batch 65536 --steps 30 --warmup 10 --no-parity | head -6
bash tools_prof.sh c16_b65536_zipf --batch 65536 --dist zipf --steps 30 --warmup 10 --no-parity | head -6
