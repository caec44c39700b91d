#!/bin/bash
# round 3, GPU call 24: in-step row Adam with the [m|v] record of a row as ONE 128-byte request (lane octets, DPP exchange)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c24
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py -q -m gpu -k "rows_in_step or in_step" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-220 | head
for pr in 1 0; do
  DT_ROWS_PAIR128=$pr timeout 400 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep "^{" > ${O}_line_pair$pr.json
  python -c "import sys,json; j=json.loads(open('${O}_line_pair$pr.json').read()); print('DeepFM pair128=$pr', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
done
DT_ROWS_PAIR128=1 timeout 400 python bench.py --model DCN --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep "^{" > ${O}_line_dcn.json
python -c "import sys,json; j=json.loads(open('${O}_line_dcn.json').read()); print('DCN', round(j['value']/1e6,3), j['step_us']['median'], (j.get('parity') or {}).get('ok'))"
DT_ROWS_PAIR128=1 timeout 400 python bench.py --dist zipf --no-cpu-baseline --no-parity --steps 200 --warmup 20 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('zipf', round(j['value']/1e6,3), j['step_us']['median'])"
