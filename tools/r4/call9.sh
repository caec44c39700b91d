#!/bin/bash
# round 4, call 9: CIN on split-bf16 matrix cores: layer tests, xDeepFM at B = 8192 vs the oracle, bench A/B, kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c9
O=gpurun_out/r4c9
timeout 600 python -m pytest tests/test_x3_gpu.py tests/test_bf16_gpu.py -q -s -k "cin or xdeepfm or bf16" 2>&1 | tail -40 > $O/t_cin.txt
python bench.py --model xDeepFM --cin bf16x3 --no-cpu-baseline --steps 40 --warmup 10 > $O/line_x3.json 2> $O/line_x3.err
python bench.py --model xDeepFM --cin f32 --no-cpu-baseline --no-parity --steps 40 --warmup 10 > $O/line_f32.json 2> $O/line_f32.err
python bench.py --model xDeepFM --cin bf16 --no-cpu-baseline --no-parity --steps 40 --warmup 10 > $O/line_bf16.json 2> $O/line_bf16.err
bash tools_prof.sh r4c9_x3 --model xDeepFM --cin bf16x3 --steps 20 --warmup 5 --no-parity > $O/stats_x3.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
tail -n 14 $O/t_cin.txt | cut -c1-300; head -14 $O/stats_x3.txt
for f in x3 f32 bf16 driver; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,3),'M rows/s', round(j['ms_per_step']*1e3,1),'us', round(j['step_us']['median'],1), j['roofline'].get('frac'), j['roofline'].get('frac_of_f32_mfma_peak'), p.get('ok'), {k:(p.get('uniform') or {}).get(k) for k in ('max_abs_logit_err','dense_grad_rel_err','dense_grad_l2_rel_err','rows_grad_rel_err','relu_units_near_kink')})
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
