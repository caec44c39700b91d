#!/bin/bash
# round 5, call 8: DCN's cross vectors summed over the tiles by the weight-gradient launch's matrix waves
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_compiled_gpu.py tests/test_parallel_gpu.py -m gpu -x -q -k "dcn or DCN" > gpurun_out/c8_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/c8_tests.txt | cut -c1-300
line() { grep "^{" $1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$2', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'parity', (j.get('parity') or {}).get('ok'))" || tail -5 ${1%.json}.err; }
timeout 300 python bench.py --model DCN --no-cpu-baseline --no-extras > gpurun_out/c8_dcn.json 2> gpurun_out/c8_dcn.err; line gpurun_out/c8_dcn.json dcn
DT_AMD_CHAIN=0 timeout 300 python bench.py --model DCN --no-cpu-baseline --no-extras --no-parity > gpurun_out/c8_dcn_nochain.json 2> gpurun_out/c8_dcn_nochain.err; line gpurun_out/c8_dcn_nochain.json dcn_nochain
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-parity > gpurun_out/c8_deepfm.json 2> gpurun_out/c8_deepfm.err; line gpurun_out/c8_deepfm.json deepfm
bash tools_prof.sh c8_dcn --model DCN --steps 100 --warmup 10 --no-parity | head -7
