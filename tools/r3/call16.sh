#!/bin/bash
# round 3, GPU call 16: CIN weight gradient through per-split slabs + one reduction instead of 16.7 M float atomics per layer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c16
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_reference_models_gpu.py tests/test_models_gpu.py tests/test_edge_cases_gpu.py tests/test_headline_gpu.py tests/test_f3_gpu.py -q -m gpu -k "cin or xdeepfm or fgcnn or CIN or xDeepFM" 2>&1 | grep -E "FAILED|passed|failed" | cut -c1-200 | head
timeout 600 python bench.py --model xDeepFM --no-cpu-baseline --steps 20 --warmup 3 > ${O}_line_xdeepfm.json 2> ${O}_line_xdeepfm.err
grep "^{" ${O}_line_xdeepfm.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('xdeepfm slabs', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))" || tail -5 ${O}_line_xdeepfm.err
DT_AMD_CIN_WGRAD_ATOMIC=1 timeout 600 python bench.py --model xDeepFM --no-cpu-baseline --no-parity --steps 20 --warmup 3 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('xdeepfm atomics', round(j['value']/1e6,3), j['step_us']['median'])"
timeout 400 bash tools_prof.sh r3c16_prof_xdeepfm --model xDeepFM --steps 20 --warmup 3 --no-parity > ${O}_stats.txt 2>&1
head -9 ${O}_stats.txt
