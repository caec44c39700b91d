#!/bin/bash
# round 3, GPU call 13: block order inside k_prep (election first) and k_finish_step (segments first)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c13
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_optim_gpu.py -q -m gpu -k "not xdeepfm and not autoint" 2>&1 | grep -E "^E  |Error|passed|failed|FAILED" | head
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), j['step_us'])"; done
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --dist zipf 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('zipf', round(j['value']/1e6,2), j['step_us'])"
timeout 400 bash tools_prof.sh r3c13_prof --steps 100 --warmup 10 --no-parity > ${O}_stats.txt 2>&1
head -7 ${O}_stats.txt
