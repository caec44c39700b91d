#!/bin/bash
# round 4, call 21: whole GPU suite (row election on the layer-by-layer path) + A/B of AutoInt / xDeepFM with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c21
O=gpurun_out/r4c21
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1
grep -E "passed|failed|error|FAILED" $O/pytest.log | tail -12
python bench.py --model AutoInt --steps 50 --warmup 10 --no-cpu-baseline > $O/line_autoint.json 2> $O/line_autoint.err
DT_AMD_ELECT_ROWS=0 python bench.py --model AutoInt --steps 50 --warmup 10 --no-cpu-baseline --no-parity > $O/line_autoint_noelect.json 2> $O/line_autoint_noelect.err
python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $O/line_xdeepfm.json 2> $O/line_xdeepfm.err
DT_AMD_ELECT_ROWS=0 python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $O/line_xdeepfm_noelect.json 2> $O/line_xdeepfm_noelect.err
for f in autoint autoint_noelect xdeepfm xdeepfm_noelect; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    print('$f', round(j['value']/1e6,3),'M rows/s', 'ms/step', round(j['ms_per_step'],4), 'parity', j.get('parity',{}).get('ok'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
