#!/bin/bash
# round 3, GPU call 23: row-owned tables — the gradient all-to-all behind the row-gradient launch, the step's last launch beside it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c23
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_parallel_gpu.py -q -m gpu -k "sharded or parallel or strategy" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-220 | head
for ov in 1 0; do
  DT_AMD_SHARDED_OVERLAP=$ov timeout 400 python bench.py --force-sharded --no-cpu-baseline --no-parity --steps 200 --warmup 20 2>${O}_sh$ov.err | grep "^{" > ${O}_sh$ov.json
  python -c "import sys,json; j=json.loads(open('${O}_sh$ov.json').read()); print('sharded W=1 overlap=$ov', round(j['value']/1e6,3), round(j['ms_per_step']*1e3,1), j.get('phases'))" || tail -5 ${O}_sh$ov.err
done
