#!/bin/bash
# round 6, call 25: projection weights pre-split into LDS (forward: 8-wave blocks; backward: region 0), per-channel constants in LDS: parity, lines, kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c25; mkdir -p $O
timeout 1200 python -m pytest tests/test_autoint_gpu.py tests/test_headline_gpu.py -q -m gpu -x -k "autoint or AutoInt or stacked or deferred or attention or head" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for m in "" "--attn bf16"; do
  tag=$(echo "$m" | tr -d ' -'); tag=${tag:-default}
  timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --model AutoInt --no-cpu-baseline $m > $O/autoint_$tag.json 2> $O/autoint_$tag.err
done
python - <<'PY'
import json
for f in ['default','attnbf16']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c25/autoint_{f}.json') if l.startswith('{')][-1]
        print(f'{f:10s}', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,3), 'M rows/s', j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c25/autoint_{f}.err').read()[-600:])
PY
bash tools_prof.sh r6c25_autoint --steps 100 --warmup 20 --model AutoInt --no-parity | cut -c1-160
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6c25_autoint/r6c25_autoint_kernel_stats.csv')))
for r in rows[14:24]:
    print(f"{float(r['AverageNs'])/1e3:9.2f}us x{r['Calls']:>5} {float(r['Percentage']):5.1f}%  {r['Name'][:90]}")
PY
rm -f gpurun_out/r6c25_autoint/*kernel_trace.csv
