#!/bin/bash
# round 3, GPU call 22: whole GPU suite after the AutoInt / loss-head changes; xDeepFM and AutoInt lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c22
timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > ${O}_tests.txt
grep -E "passed|failed|FAILED" ${O}_tests.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for m in AutoInt xDeepFM; do
  timeout 400 python bench.py --model $m --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep "^{" > ${O}_line_$m.json
  python -c "import sys,json; j=json.loads(open('${O}_line_$m.json').read()); print('$m', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
done
