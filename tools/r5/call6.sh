#!/bin/bash
# round 5, call 6: chained steps (step i prepares step i + 1: four launches from the second step of an execution on)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_compiled_gpu.py tests/test_fused_gpu.py tests/test_headline_gpu.py -m gpu -x -q > gpurun_out/c6_tests.txt 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/c6_tests.txt | cut -c1-400
line() { grep "^{" $1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$2', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'frac', round(j['roofline']['frac'],4), 'parity', (j.get('parity') or {}).get('ok'), 'fit', j.get('fit_rows_per_s'), 'fb', j.get('fwd_bwd_only_rows_per_s'))" || tail -5 ${1%.json}.err; }
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c6_b8192.json 2> gpurun_out/c6_b8192.err; line gpurun_out/c6_b8192.json chained
DT_AMD_CHAIN=0 timeout 300 python bench.py --no-cpu-baseline --no-parity > gpurun_out/c6_b8192_nochain.json 2> gpurun_out/c6_b8192_nochain.err; line gpurun_out/c6_b8192_nochain.json nochain
bash tools_prof.sh c6_b8192 --steps 100 --warmup 10 --no-parity | head -8
bash tools_prof.sh c6_dcn --model DCN --steps 100 --warmup 10 --no-parity | head -8
