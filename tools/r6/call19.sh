#!/bin/bash
# round 6, call 19: deferred BatchNormalization between stacked interacting layers (AiXn) + double sums read in place:
# parity tests, AutoInt lines, then the Python origins of the small torch kernels of the AutoInt / xDeepFM steps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c19; mkdir -p $O
timeout 1200 python -m pytest tests/test_autoint_gpu.py tests/test_headline_gpu.py -q -m gpu -x -k "autoint or AutoInt or stacked or deferred or attention" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for m in "" "--attn f32" "--attn bf16"; do
  tag=$(echo "$m" | tr -d ' -'); tag=${tag:-default}
  timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --model AutoInt --no-cpu-baseline $m > $O/autoint_$tag.json 2> $O/autoint_$tag.err
done
python - <<'PY'
import json
for f in ['default','attnf32','attnbf16']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c19/autoint_{f}.json') if l.startswith('{')][-1]
        print(f'{f:10s}', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,3), 'M rows/s', j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c19/autoint_{f}.err').read()[-600:])
PY
timeout 600 python tools/r6/glue_trace.py AutoInt 3 > $O/glue_autoint.txt 2>&1
timeout 600 python tools/r6/glue_trace.py xDeepFM 3 > $O/glue_xdeepfm.txt 2>&1
head -70 $O/glue_autoint.txt; echo; head -90 $O/glue_xdeepfm.txt
