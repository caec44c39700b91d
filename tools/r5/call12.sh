#!/bin/bash
# round 5, call 12: election list spill (hot rows beyond 8192 lookups of a partition) — tests + the B = 65536 lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "chained or rows_in_step" > gpurun_out/c12_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/c12_tests.txt | cut -c1-300
line() { grep "^{" $1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); p=j.get('parity') or {}; print('$2', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'parity', p.get('ok'), [ (k, (p[k].get('in_step_optimizer') or {}).get('ok'), ((p[k].get('in_step_optimizer') or {}).get('vs_oracle') or {}).get('ok')) for k in ('uniform','zipf') if k in p])" || tail -5 ${1%.json}.err; }
timeout 600 python bench.py --batch 65536 --steps 30 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/c12_b65536.json 2> gpurun_out/c12_b65536.err; line gpurun_out/c12_b65536.json b65536
timeout 600 python bench.py --batch 65536 --dist zipf --steps 30 --warmup 10 --no-cpu-baseline --no-extras --no-parity > gpurun_out/c12_b65536_zipf.json 2> gpurun_out/c12_b65536_zipf.err; line gpurun_out/c12_b65536_zipf.json b65536_zipf
timeout 600 python bench.py --batch 32768 --dist zipf --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-parity > gpurun_out/c12_b32768_zipf.json 2> gpurun_out/c12_b32768_zipf.err; line gpurun_out/c12_b32768_zipf.json b32768_zipf
