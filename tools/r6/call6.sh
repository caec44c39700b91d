#!/bin/bash
# round 6, call 6: fresh-box driver command (#5); attention modes: tests, AutoInt lines f32 / bf16x2 / bf16
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c6; mkdir -p $O gpurun_out/r6drv
sleep 8
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6drv/line_5.json 2> gpurun_out/r6drv/line_5.err
timeout 900 python -m pytest tests/test_autoint_gpu.py -q -m gpu > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for a in f32 bf16x2 bf16; do
  timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --model AutoInt --attn $a > $O/autoint_$a.json 2> $O/autoint_$a.err
done
python - <<'PY'
import json,glob
for f in ['gpurun_out/r6drv/line_5.json']+sorted(glob.glob('gpurun_out/r6c6/*.json')):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        su=j['step_us']
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), 'M; rep', su.get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), j['dtype'])
        p=j.get('parity')
        if p:
            u=p.get('uniform') or p.get('zipf') or {}
            print('   ', {k:u[k] for k in u if 'err' in k})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
