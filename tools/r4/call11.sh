#!/bin/bash
# round 4, call 11: tower_x3 tuning (deeper B prefetch, dXn's W1 rows requested three phases earlier); dt_graph_upload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c11
O=gpurun_out/r4c11
timeout 300 python -m pytest tests/test_x3_gpu.py tests/test_compiled_gpu.py -q -x -k "not cin and not xdeepfm" 2>&1 | tail -4 > $O/t_a.txt
python bench.py --no-cpu-baseline > $O/line_default.json 2> $O/line_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
bash tools_prof.sh r4c11_deepfm --steps 100 --warmup 10 --no-parity > $O/stats.txt 2>&1
ROWS=1 DT_DEEPFM_STAMPS=1 timeout 100 python tools/phase_times.py > $O/stamps.txt 2>&1
tail -n 2 $O/t_a.txt; head -7 $O/stats.txt; head -16 $O/stamps.txt
for f in default driver; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', round(j['step_us']['median'],1), j.get('first_replay_us'), j['config'].get('graph_uploaded_before_first_replay'), j.get('fit_rows_per_s'), p.get('ok'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
