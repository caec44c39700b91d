#!/bin/bash
# round 4, call 5: tower_x3 v2 (six-product forward, three-product backward): parity tests, the crash of call 3's t_misc, A/B, stamps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c5
O=gpurun_out/r4c5
timeout 300 python -X faulthandler -m pytest tests/test_compiled_gpu.py tests/test_weights_gpu.py -q -x > $O/t_misc_full.txt 2>&1
grep -n "Fatal\|Error\|passed\|failed" $O/t_misc_full.txt | head -20; grep -n -B2 -A25 "Fatal Python" $O/t_misc_full.txt | cut -c1-300 | head -60
timeout 600 python -m pytest tests/test_x3_gpu.py -q -s 2>&1 | tail -80 > $O/t_x3.txt
python bench.py --no-cpu-baseline --tower bf16x3 > $O/line_x3.json 2> $O/line_x3.err
python bench.py --gpus 1 --steps 20 --warmup 5 --tower bf16x3 > $O/line_x3_driver.json 2> $O/line_x3_driver.err
bash tools_prof.sh r4c5_x3 --steps 100 --warmup 10 --no-parity --tower bf16x3 > $O/stats_x3.txt 2>&1
DT_AMD_TOWER_DTYPE=bf16x3 ROWS=1 DT_DEEPFM_STAMPS=1 timeout 100 python tools/phase_times.py > $O/stamps_x3.txt 2>&1
tail -n 12 $O/t_x3.txt | cut -c1-400; head -8 $O/stats_x3.txt; head -16 $O/stamps_x3.txt
for f in x3 x3_driver; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', j['step_us']['median'], j.get('first_replay_us'), j.get('fit_rows_per_s'), j.get('fwd_bwd_only_rows_per_s'), p.get('ok'))
    for k in ('uniform','zipf'):
        d=p.get(k) or {}
        print('  ',k,{a:d.get(a) for a in ('max_abs_logit_err','dense_grad_rel_err','rows_grad_rel_err','relu_kink_retries','relu_units_near_kink','adam_rows_rel_err')}, (d.get('in_step_optimizer') or {}).get('vs_oracle',{}).get('ok'), {a:(d.get('in_step_optimizer') or {}).get('vs_oracle',{}).get(a) for a in ('rows_p_err','dense_p_err','rows_m_err','dense_m_err')})
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
