#!/bin/bash
# round 3, GPU call 33: bf16 CIN backward overwrites grad_x0 / grad_xk (no 67 MB zero fills)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c33
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_reference_models_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "bf16 or cin" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-200 | head
DT_AMD_CIN_DTYPE=bf16 timeout 600 python bench.py --model xDeepFM --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | grep "^{" > ${O}_line.json
python -c "import sys,json; j=json.loads(open('${O}_line.json').read()); print('xdeepfm bf16', round(j['value']/1e6,4), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
DT_AMD_CIN_DTYPE=bf16 timeout 400 bash tools_prof.sh r3c33_prof --model xDeepFM --steps 20 --warmup 3 --no-parity 2>&1 | head -8 | cut -c1-130
