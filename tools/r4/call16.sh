#!/bin/bash
# round 4, call 16: DCN through k_tower_x3<N, kCrossMax> (the flag now reaches dt_dcn_train_step) + two-level gather tickets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c16
O=gpurun_out/r4c16
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_compiled_gpu.py tests/test_feed.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_parallel_gpu.py -m gpu -x -q -k "dcn or DCN" > $O/pytest_headline.log 2>&1
tail -3 $O/pytest_headline.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/line_driver.json 2> $O/line_driver.err
python bench.py --gpus 1 --no-cpu-baseline > $O/line_default.json 2> $O/line_default.err
python bench.py --gpus 1 --model DCN --no-cpu-baseline > $O/line_dcn.json 2> $O/line_dcn.err
for f in driver default dcn; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    s=j['step_us']
    print('$f', round(j['value']/1e6,2),'M rows/s', 'wall', round(s['wall_us']), 'gpu', round(s['gpu_us']), 'median', round(s['median'],1), 'parity', j.get('parity',{}).get('ok'))
    p=j.get('parity',{}); print('   ', {k: p[k] for k in p if 'err' in k})
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
bash tools_prof.sh r4c16_dcn --model DCN --steps 100 --warmup 10 --no-parity > $O/stats_dcn.txt 2>&1
head -9 $O/stats_dcn.txt
