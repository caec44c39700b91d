#!/bin/bash
# round 6, call 23: the compiled loop keeps the layer-by-layer steps' (loss, logits) by reference (no copy launches): tests + lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c23; mkdir -p $O
timeout 1500 python -m pytest tests/test_compiled_gpu.py tests/test_autoint_gpu.py tests/test_headline_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --model AutoInt --no-cpu-baseline > $O/autoint.json 2> $O/autoint.err
timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --model AutoInt --no-cpu-baseline --attn bf16 > $O/autoint_bf16.json 2> $O/autoint_bf16.err
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 20 --model xDeepFM --no-cpu-baseline > $O/xdeepfm.json 2> $O/xdeepfm.err
python - <<'PY'
import json
for f in ['autoint','autoint_bf16','xdeepfm']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c23/{f}.json') if l.startswith('{')][-1]
        print(f'{f:12s}', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,3), 'M rows/s', j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), 'fit', j.get('fit_rows_per_s'))
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c23/{f}.err').read()[-600:])
PY
