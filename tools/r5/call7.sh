#!/bin/bash
# round 5, call 7: DCN's cross vectors back in per-tile records (summed by the finishing launch); the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c7_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/c7_tests.txt | cut -c1-300
bash tools_prof.sh c7_dcn --model DCN --steps 100 --warmup 10 | head -8
grep '"metric"' gpurun_out/c7_dcn.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('dcn', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'parity', (j.get('parity') or {}).get('ok'))"
