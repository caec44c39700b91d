#!/bin/bash
# round 5, call 2: the chunked election (any batch size) — the fused / headline / compiled suites, then the B = 8192 line and
# SURVEY 8(d)(ii)'s large-batch lines with their kernel tables
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_compiled_gpu.py tests/test_optim_gpu.py -m gpu -x -q > gpurun_out/c2_tests.txt 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/c2_tests.txt
line() { grep "^{" $1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$2', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'frac', round(j['roofline']['frac'],4), 'parity', (j.get('parity') or {}).get('ok'), 'fit', j.get('fit_rows_per_s'), 'fb', j.get('fwd_bwd_only_rows_per_s'))" || tail -5 ${1%.json}.err; }
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c2_b8192.json 2> gpurun_out/c2_b8192.err; line gpurun_out/c2_b8192.json b8192
timeout 400 python bench.py --batch 32768 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/c2_b32768.json 2> gpurun_out/c2_b32768.err; line gpurun_out/c2_b32768.json b32768
timeout 500 python bench.py --batch 65536 --steps 30 --warmup 10 --steps-per-graph 5 --no-cpu-baseline > gpurun_out/c2_b65536.json 2> gpurun_out/c2_b65536.err; line gpurun_out/c2_b65536.json b65536
bash tools_prof.sh c2_b32768 --batch 32768 --steps 50 --warmup 10 --no-parity
bash tools_prof.sh c2_b65536 --batch 65536 --steps 30 --warmup 10 --steps-per-graph 5 --no-parity
