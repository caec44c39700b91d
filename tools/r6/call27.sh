#!/bin/bash
# round 6, call 27: the hot-row walk of the finishing launch with the next chunk's member ids one round trip ahead: parity, Zipf / uniform lines, F under the tracer
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c27; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_optim_gpu.py tests/test_headline_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dist zipf > $O/zipf.json 2> $O/zipf.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras > $O/uni.json 2> $O/uni.err
timeout 600 python bench.py --gpus 1 --steps 30 --warmup 10 --batch 65536 --no-cpu-baseline --dist zipf > $O/zipf65536.json 2> $O/zipf65536.err
python - <<'PY'
import json
for f in ['zipf','uni','zipf65536']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c27/{f}.json') if l.startswith('{')][-1]
        print(f'{f:10s}', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c27/{f}.err').read()[-500:])
PY
bash tools_prof.sh r6c27_zipf --steps 100 --warmup 10 --no-parity --dist zipf | head -6 | cut -c1-120
rm -f gpurun_out/r6c27_zipf/*kernel_trace.csv
