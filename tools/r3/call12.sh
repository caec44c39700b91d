#!/bin/bash
# round 3, GPU call 12: write-through (sc1) stores for what the next kernel reads — does the kernel boundary get cheaper?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c12
for wt in 0 1 2 3; do
  DT_WT=$wt timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 2>/dev/null | grep "^{" > ${O}_line_wt$wt.json
  python - <<PY
import json
j=json.loads(open('${O}_line_wt$wt.json').read())
print('DT_WT=$wt', round(j['value']/1e6,2), j['step_us'])
PY
done
DT_WT=3 timeout 400 bash tools_prof.sh r3c12_prof_wt3 --steps 100 --warmup 10 --no-parity > ${O}_stats_wt3.txt 2>&1
head -7 ${O}_stats_wt3.txt
DT_WT=0 timeout 400 bash tools_prof.sh r3c12_prof_wt0 --steps 100 --warmup 10 --no-parity > ${O}_stats_wt0.txt 2>&1
head -7 ${O}_stats_wt0.txt
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py -q -m gpu -k "not xdeepfm and not autoint" 2>&1 | grep -E "^E  |Error|passed|failed|FAILED" | head
