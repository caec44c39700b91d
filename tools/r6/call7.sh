#!/bin/bash
# round 6, call 7: linked BatchNormalization-backward sums of stacked attention layers; AutoInt default (split-bf16) lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c7; mkdir -p $O
timeout 900 python -m pytest tests/test_autoint_gpu.py tests/test_bf16_gpu.py -q -m gpu > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_reference_models_gpu.py tests/test_models_gpu.py tests/test_golden_gpu.py -q -m gpu -x > $O/pytest2.txt 2>&1
tail -4 $O/pytest2.txt
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --model AutoInt > $O/autoint_default.json 2> $O/autoint_default.err
DT_AMD_AUTOINT_LINK=0 timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-parity --model AutoInt > $O/autoint_nolink.json 2> $O/autoint_nolink.err
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --model AutoInt --attn bf16 > $O/autoint_bf16.json 2> $O/autoint_bf16.err
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-parity --model AutoInt --attn f32 > $O/autoint_f32.json 2> $O/autoint_f32.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c7/*.json')):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        su=j['step_us']
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), 'M; rep', su.get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), j['dtype'])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
