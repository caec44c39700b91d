#!/bin/bash
# round 4, call 20: the whole GPU suite (plain-bf16 tower mode, explicit slot stride, CIN dgrad back to one chunk of lookahead)
# + the bf16-tower bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c20
O=gpurun_out/r4c20
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1
grep -E "passed|failed|error|FAILED" $O/pytest.log | tail -6
grep "bf16 tower" $O/pytest.log | head
python bench.py --tower bf16 --no-cpu-baseline > $O/line_deepfm_bf16.json 2> $O/line_deepfm_bf16.err
python bench.py --model DCN --tower bf16 --no-cpu-baseline > $O/line_dcn_bf16.json 2> $O/line_dcn_bf16.err
for f in deepfm_bf16 dcn_bf16; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity',{})
    print('$f', round(j['value']/1e6,2),'M rows/s', 'median', round(j['step_us']['median'],1), 'parity', p.get('ok'), j['config'].get('tower_mfma'))
    u=p.get('uniform',{}); print('   ', {k:u.get(k) for k in ('max_abs_logit_err','max_abs_logit','dense_grad_rel_err','dense_grad_l2_rel_err','rows_grad_rel_err','rows_grad_l2_rel_err')})
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
