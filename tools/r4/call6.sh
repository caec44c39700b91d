#!/bin/bash
# round 4, call 6: split-bf16 tower as the default of the fused DeepFM step; device-cursor feed; the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c6
O=gpurun_out/r4c6
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -60 > $O/t_all.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
python bench.py --no-cpu-baseline > $O/line_default.json 2> $O/line_default.err
python bench.py --no-cpu-baseline --dist zipf --no-parity > $O/line_zipf.json 2> $O/line_zipf.err
grep -n "passed\|failed\|FAILED" $O/t_all.txt | head -30
for f in driver default zipf; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', j['step_us']['median'], j.get('first_replay_us'), j.get('fit_rows_per_s'), j.get('fwd_bwd_only_rows_per_s'), p.get('ok'), j['config'].get('tower_mfma'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
