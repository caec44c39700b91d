#!/bin/bash
# round 3, GPU call 25: rows per pass of the row-gradient / row-Adam epilogue (kRowsUB 4 -> 6: fewer dependent round trips per wave)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c25
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py -q -m gpu -k "rows_in_step or in_step or dcn or DCN" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-220 | head
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --no-parity --steps 200 --warmup 20 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('DeepFM UB6', round(j['value']/1e6,3), j['step_us']['median'])"
done
timeout 400 python bench.py --model DCN --no-cpu-baseline --no-parity --steps 200 --warmup 20 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('DCN UB6', round(j['value']/1e6,3), j['step_us']['median'])"
timeout 400 python bench.py --dist zipf --no-cpu-baseline --no-parity --steps 200 --warmup 20 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('zipf UB6', round(j['value']/1e6,3), j['step_us']['median'])"
timeout 300 bash tools_prof.sh r3c25_prof --steps 100 --warmup 10 --no-parity 2>&1 | head -7 | cut -c1-120
