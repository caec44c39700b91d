#!/bin/bash
# round 5, call 1: the whole GPU suite on the five-launch step (record / BN batch sums as double atomics, no reduction launch,
# no BN level-1 blocks), the bench line, the kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/c1_tests.txt
timeout 400 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
grep "^{" gpurun_out/c1_bench.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'frac', round(j['roofline']['frac'],4), 'parity', (j.get('parity') or {}).get('ok'), 'fit', j.get('fit_rows_per_s'), 'fb', j.get('fwd_bwd_only_rows_per_s'))" || tail -5 gpurun_out/c1_bench.err
bash tools_prof.sh c1_deepfm --steps 100 --warmup 10 --no-parity
