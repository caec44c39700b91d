#!/bin/bash
# round 3, GPU call 20: AutoInt BN statistics in the attention kernel epilogue (dt_autoint_fwd_bn: 2 launches instead of 4)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c20
timeout 900 python -m pytest tests/test_autoint_gpu.py tests/test_models_gpu.py -q -m gpu -k "autoint or AutoInt" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-220 | head
timeout 400 bash tools_prof.sh r3c20_prof_autoint --model AutoInt --steps 20 --warmup 3 --no-parity > ${O}_stats.txt 2>&1
head -8 ${O}_stats.txt | cut -c1-160
timeout 400 python bench.py --model AutoInt --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep "^{" > ${O}_line.json
python -c "import sys,json; j=json.loads(open('${O}_line.json').read()); print('autoint', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
