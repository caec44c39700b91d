#!/bin/bash
# round 3, GPU call 17: sample weights + dense-input dropout inside the fused DeepFM / DCN step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c17
timeout 900 python -m pytest tests/test_fused_gpu.py -q -m gpu -x 2>&1 | tail -15 > ${O}_tests.txt
grep -E "passed|failed" ${O}_tests.txt
for m in DeepFM DCN; do
  timeout 400 python bench.py --model $m --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep "^{" > ${O}_line_$m.json
  python -c "import sys,json; j=json.loads(open('${O}_line_$m.json').read()); print('$m', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
done
