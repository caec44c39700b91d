#!/bin/bash
# round 3, GPU call 10: bucketed sparse exchange (dt_rows_compact), data-parallel step structure at world size 1, two-process tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c10
timeout 1200 python -m pytest tests/test_optim_gpu.py tests/test_parallel_gpu.py tests/test_fused_gpu.py -q -m gpu 2>&1 | grep -E "^E  |AssertionError|Error|passed|failed" | head -30 > ${O}_tests.txt
cat ${O}_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --force-dp > ${O}_line_dp1.json 2> ${O}_line_dp1.err
DT_AMD_DP_DEDUPE=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --force-dp > ${O}_line_dp1_nodedupe.json 2> ${O}_line_dp1_nodedupe.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --force-dp --dist zipf --bucket-ratio 0.6 > ${O}_line_dp1_zipf.json 2> ${O}_line_dp1_zipf.err
DT_AMD_DP_DEDUPE=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --force-dp --dist zipf > ${O}_line_dp1_zipf_nodedupe.json 2> ${O}_line_dp1_zipf_nodedupe.err
for f in dp1 dp1_nodedupe dp1_zipf dp1_zipf_nodedupe; do echo $f; python - <<PY
import json
try:
    j=json.loads(open('${O}_line_$f.json').read().strip().splitlines()[-1])
    print(j['value'], j['step_us']['median'], j.get('phases'), j['config'].get('sparse_exchange'))
except Exception as e:
    print('ERR', e); print(open('${O}_line_$f.err').read()[-1500:])
PY
done
