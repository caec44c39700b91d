#!/bin/bash
# round 3, GPU call 15: CIN fp32 kernels with 16-byte operand reads (K-permuted MFMA steps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c15
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_reference_models_gpu.py tests/test_models_gpu.py tests/test_edge_cases_gpu.py -q -m gpu -k "cin or xdeepfm or fgcnn or CIN or xDeepFM" 2>&1 | grep -E "^E  |Error|passed|failed|FAILED" | head -20
timeout 600 python bench.py --model xDeepFM --no-cpu-baseline --steps 20 --warmup 3 > ${O}_line_xdeepfm.json 2> ${O}_line_xdeepfm.err
grep "^{" ${O}_line_xdeepfm.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('xdeepfm', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))" || tail -5 ${O}_line_xdeepfm.err
timeout 400 bash tools_prof.sh r3c15_prof_xdeepfm --model xDeepFM --steps 20 --warmup 3 --no-parity > ${O}_stats.txt 2>&1
head -8 ${O}_stats.txt
