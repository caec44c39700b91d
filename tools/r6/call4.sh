#!/bin/bash
# round 6, call 4: fresh-box driver command (#2), then the bf16 attention mode (tests + bench lines) and the DP compiled tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c4; mkdir -p $O gpurun_out/r6drv
sleep 8
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6drv/line_2.json 2> gpurun_out/r6drv/line_2.err
timeout 900 python -m pytest tests/test_autoint_gpu.py tests/test_compiled_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --model AutoInt > $O/autoint_f32.json 2> $O/autoint_f32.err
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --model AutoInt --attn bf16 > $O/autoint_bf16.json 2> $O/autoint_bf16.err
python - <<'PY'
import json,glob
for f in ['gpurun_out/r6drv/line_2.json']+sorted(glob.glob('gpurun_out/r6c4/*.json')):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        su=j['step_us']
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), 'M; rep', su.get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), 'roofline', {k:j['roofline'].get(k) for k in ('frac','frac_of_f32_mfma_peak')}, 'dtype', j['dtype'])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
