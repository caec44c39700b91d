#!/bin/bash
# round 6, call 17: the next step's election inside the finishing launch (4096-slot table): parity tests, then lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c17; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_compiled_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver.json 2> $O/driver.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/deepfm200.json 2> $O/deepfm200.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dist zipf > $O/zipf200.json 2> $O/zipf200.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --model DCN > $O/dcn200.json 2> $O/dcn200.err
python - <<'PY'
import json
for f in ['driver','deepfm200','zipf200','dcn200']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c17/{f}.json') if l.startswith('{')][-1]
        print(f'{f:12s}', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c17/{f}.err').read()[-400:])
PY
