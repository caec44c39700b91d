#!/bin/bash
# round 3, GPU call 26: CIN split + pool as one kernel each way (torch's strided reduce ran at 1 TB/s)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c26
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_reference_models_gpu.py tests/test_models_gpu.py tests/test_headline_gpu.py tests/test_bf16_gpu.py -q -m gpu -k "cin or xdeepfm or CIN or xDeepFM" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-200 | head
timeout 600 python bench.py --model xDeepFM --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | grep "^{" > ${O}_line_xdeepfm.json
python -c "import sys,json; j=json.loads(open('${O}_line_xdeepfm.json').read()); print('xdeepfm', round(j['value']/1e6,4), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
timeout 400 bash tools_prof.sh r3c26_prof_xdeepfm --model xDeepFM --steps 20 --warmup 3 --no-parity 2>&1 | head -12 | cut -c1-140
