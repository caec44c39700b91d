#!/bin/bash
# round 3, GPU call 11: DCN on the pipelined step (cross term inside the dXn GEMM, in-step optimizer)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c11
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_optim_gpu.py tests/test_headline_gpu.py tests/test_reference_models_gpu.py -q -m gpu -k "not xdeepfm and not autoint" 2>&1 | grep -E "^E  |AssertionError|Error|passed|failed|FAILED" | head -30 > ${O}_tests.txt
cat ${O}_tests.txt
timeout 300 python bench.py --model DCN --no-cpu-baseline --steps 200 > ${O}_line_dcn.json 2> ${O}_line_dcn.err
DT_STEP_PIPE=0 timeout 300 python bench.py --model DCN --no-cpu-baseline --no-parity --no-extras --steps 200 > ${O}_line_dcn_old.json 2> ${O}_line_dcn_old.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 > ${O}_line_deepfm.json 2> ${O}_line_deepfm.err
for f in dcn dcn_old deepfm; do python - <<PY
import json
for l in open('${O}_line_$f.json'):
    if l.startswith('{'):
        j=json.loads(l); print('$f', round(j['value']/1e6,2), 'M rows/s', j['step_us'], (j.get('parity') or {}).get('ok'))
PY
tail -2 ${O}_line_$f.err; done
timeout 400 bash tools_prof.sh r3c11_prof_dcn --model DCN --steps 100 --warmup 10 --no-parity > ${O}_stats_dcn.txt 2>&1
head -8 ${O}_stats_dcn.txt
