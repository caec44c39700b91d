#!/bin/bash
# round 4, call 3: the split-bf16 tower: parity tests, A/B bench on one box, kernel stats, phase stamps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c3
O=gpurun_out/r4c3
timeout 600 python -m pytest tests/test_x3_gpu.py -q -s 2>&1 | tail -40 > $O/t_x3.txt
python bench.py --no-cpu-baseline --tower f32 > $O/line_f32.json 2> $O/line_f32.err
python bench.py --no-cpu-baseline --tower bf16x3 > $O/line_x3.json 2> $O/line_x3.err
python bench.py --gpus 1 --steps 20 --warmup 5 --tower bf16x3 > $O/line_x3_driver.json 2> $O/line_x3_driver.err
bash tools_prof.sh r4c3_x3 --steps 100 --warmup 10 --no-parity --tower bf16x3 > $O/stats_x3.txt 2>&1
DT_AMD_TOWER_DTYPE=bf16x3 ROWS=1 DT_DEEPFM_STAMPS=1 timeout 100 python tools/phase_times.py > $O/stamps_x3.txt 2>&1
timeout 300 python -m pytest tests/test_weights_gpu.py tests/test_compiled_gpu.py -q 2>&1 | tail -5 > $O/t_misc.txt
tail -n 12 $O/t_x3.txt; tail -n 3 $O/t_misc.txt; head -12 $O/stats_x3.txt; head -16 $O/stamps_x3.txt
for f in f32 x3 x3_driver; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', j['step_us']['median'], j.get('first_replay_us'), j.get('fit_rows_per_s'), j.get('fwd_bwd_only_rows_per_s'), p.get('ok'), (p.get('uniform') or {}).get('max_abs_logit_err'), (p.get('uniform') or {}).get('dense_grad_rel_err'), (p.get('uniform') or {}).get('rows_grad_rel_err'), (p.get('zipf') or {}).get('max_abs_logit_err'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
