#!/bin/bash
# round 6, call 19/20: deferred BatchNormalization (AiXn), the Dense(1) head on the pending normalisation (rank-one gradient):
# parity tests, AutoInt lines, then the Python origins of the small torch kernels of the AutoInt / xDeepFM steps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c20; mkdir -p $O
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
        j=[json.loads(l) for l in open(f'gpurun_out/r6c20/autoint_{f}.json') if l.startswith('{')][-1]
        print(f'{f:10s}', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,3), 'M rows/s', j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c20/autoint_{f}.err').read()[-600:])
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o autoint -- python bench.py --gpus 1 --steps 100 --warmup 20 --model AutoInt --no-cpu-baseline --no-parity --no-extras > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/autoint_kernel_stats.csv && head -24 $O/autoint_kernel_stats.csv | cut -c1-150
rm -rf $O/prof
