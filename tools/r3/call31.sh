#!/bin/bash
# round 3, GPU call 31: per-stream partial buffers, cin_split_pool requires device tensors — AutoInt / xDeepFM tests and lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_autoint_gpu.py tests/test_kernels_gpu.py tests/test_models_gpu.py -q -m gpu -k "autoint or AutoInt or cin or xdeepfm" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-200 | head
for m in AutoInt xDeepFM; do
timeout 400 python bench.py --model $m --no-cpu-baseline --no-parity --steps 20 --warmup 5 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$m', round(j['value']/1e6,3), j['step_us']['median'])"
done
