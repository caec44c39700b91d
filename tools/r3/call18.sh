#!/bin/bash
# round 3, GPU call 18: AutoInt weight gradients inside k_autoint_bwd (dY never written, no Dense wgrad launch)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c18
timeout 900 python -m pytest tests/test_autoint_gpu.py tests/test_models_gpu.py tests/test_reference_models_gpu.py -q -m gpu -k "autoint or AutoInt" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-220 | head
for mode in fused dense; do
  DT_AMD_AUTOINT_WGRAD=$mode timeout 400 python bench.py --model AutoInt --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep "^{" > ${O}_line_$mode.json
  python -c "import sys,json; j=json.loads(open('${O}_line_$mode.json').read()); print('$mode', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
done
timeout 400 bash tools_prof.sh r3c18_prof_autoint --model AutoInt --steps 20 --warmup 3 --no-parity > ${O}_stats.txt 2>&1
head -14 ${O}_stats.txt | cut -c1-200
