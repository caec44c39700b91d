#!/bin/bash
# round 4, call 8: wave-per-row feed gather, pre-election as an opt-in (its test), driver + default lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c8
O=gpurun_out/r4c8
timeout 300 python -m pytest tests/test_compiled_gpu.py -q 2>&1 | tail -8 > $O/t_a.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
python bench.py --no-cpu-baseline > $O/line_default.json 2> $O/line_default.err
bash tools_prof.sh r4c8_deepfm --steps 100 --warmup 10 --no-parity > $O/stats.txt 2>&1
tail -n 3 $O/t_a.txt; head -12 $O/stats.txt
for f in driver default; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', round(j['step_us']['median'],1), j.get('first_replay_us'), j.get('fit_rows_per_s'), j.get('fwd_bwd_only_rows_per_s'), p.get('ok'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
