#!/bin/bash
# round 4, call 17: CIN kernels with 256-row blocks (forward, dgrad) and the wide wgrad blocks: parity, then xDeepFM timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c17
O=gpurun_out/r4c17
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_bf16_gpu.py -m gpu -x -q -k "cin or xdeepfm or bf16" > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline > $O/line_xdeepfm.json 2> $O/line_xdeepfm.err
DT_CIN_WIDE=0 DT_CIN_WGRAD_WIDE=0 python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $O/line_xdeepfm_narrow.json 2> $O/line_xdeepfm_narrow.err
python bench.py --model xDeepFM --cin bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $O/line_xdeepfm_bf16.json 2> $O/line_xdeepfm_bf16.err
for f in xdeepfm xdeepfm_narrow xdeepfm_bf16; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    s=j['step_us']
    print('$f', round(j['value']/1e6,3),'M rows/s', 'ms/step', round(j['ms_per_step'],3), 'parity', j.get('parity',{}).get('ok'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
bash tools_prof.sh r4c17_xdeepfm --model xDeepFM --steps 20 --warmup 5 --no-parity > $O/stats_xdeepfm.txt 2>&1
head -12 $O/stats_xdeepfm.txt
