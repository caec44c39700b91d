#!/bin/bash
# round 6, call 2: the merged launch (F || A): parity tests first, then merged vs chained bench lines (uniform, zipf)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c2; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "merged or chained" > $O/pytest_merged.txt 2>&1
tail -5 $O/pytest_merged.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/merged.json 2> $O/merged.err
DT_AMD_MERGE=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/chained.json 2> $O/chained.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras > $O/merged200.json 2> $O/merged200.err
DT_AMD_MERGE=0 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras > $O/chained200.json 2> $O/chained200.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras --dist zipf > $O/merged200_zipf.json 2> $O/merged200_zipf.err
DT_AMD_MERGE=0 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras --dist zipf > $O/chained200_zipf.json 2> $O/chained200_zipf.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras --model DCN > $O/merged200_dcn.json 2> $O/merged200_dcn.err
timeout 900 python -m pytest tests/test_compiled_gpu.py tests/test_headline_gpu.py -x -q -m gpu > $O/pytest_compiled.txt 2>&1
tail -5 $O/pytest_compiled.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c2/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        su=j['step_us']
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us  rep', su.get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), 'split', j.get('kernel_split_us'), 'clk', (su.get('clocks') or {}).get('contract_region'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
