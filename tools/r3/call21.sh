#!/bin/bash
# round 3, GPU call 21: coalesced partial reductions (AutoInt weight gradients, BN sums); bench --tables auto
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c21
timeout 900 python -m pytest tests/test_autoint_gpu.py tests/test_models_gpu.py -q -m gpu -k "autoint or AutoInt" 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-220 | head
timeout 400 bash tools_prof.sh r3c21_prof_autoint --model AutoInt --steps 20 --warmup 3 --no-parity > ${O}_stats.txt 2>&1
head -9 ${O}_stats.txt | cut -c1-160
timeout 400 python bench.py --model AutoInt --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep "^{" > ${O}_line.json
python -c "import sys,json; j=json.loads(open('${O}_line.json').read()); print('autoint', round(j['value']/1e6,3), j['step_us']['median'], j['roofline']['frac'], (j.get('parity') or {}).get('ok'))"
timeout 400 python bench.py --force-sharded --no-cpu-baseline --no-parity --steps 200 --warmup 20 2>${O}_sh.err | grep "^{" > ${O}_sh.json
python -c "import sys,json; j=json.loads(open('${O}_sh.json').read()); print('sharded W=1', round(j['value']/1e6,3), j['ms_per_step'], j.get('phases'), j['config']['parallelism'])" || tail -5 ${O}_sh.err
