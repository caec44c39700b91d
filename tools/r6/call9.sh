#!/bin/bash
# round 6, call 9: the weight-gradient GEMMs of E||D on split-bf16 matrix cores: parity (fused / headline / compiled tests), lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c9; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_compiled_gpu.py tests/test_x3_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver.json 2> $O/driver.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/deepfm200.json 2> $O/deepfm200.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --model DCN > $O/dcn200.json 2> $O/dcn200.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --tower bf16 > $O/bf16tower.json 2> $O/bf16tower.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --dist zipf > $O/zipf.json 2> $O/zipf.err
timeout 600 python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --batch 65536 > $O/b65536.json 2> $O/b65536.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c9/*.json')):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        p=j.get('parity') or {}
        u=p.get('uniform') or {}
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), 'M', j['step_us'].get('repeat_step_us'), 'parity', p.get('ok'), {k:u.get(k) for k in ('dense_grad_rel_err','rows_grad_rel_err','max_abs_logit_err')}, {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
