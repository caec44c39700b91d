#!/bin/bash
# round 4, call 14: the Cross network inside the split-bf16 tile kernel (k_tower_x3<N, kCrossMax>) + the feed gather that
# advances its own cursor: parity, then the timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c14
O=gpurun_out/r4c14
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_compiled_gpu.py tests/test_feed.py tests/test_x3_gpu.py -m gpu -x -q -k "not xdeepfm and not cin_layer" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -k "dcn or DCN" > $O/pytest_headline.log 2>&1
tail -3 $O/pytest_headline.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
python bench.py --gpus 1 --no-cpu-baseline > $O/line_default.json 2> $O/line_default.err
python bench.py --gpus 1 --model DCN --no-cpu-baseline > $O/line_dcn.json 2> $O/line_dcn.err
python bench.py --gpus 1 --model DCN --tower f32 --no-cpu-baseline --no-parity > $O/line_dcn_f32.json 2> $O/line_dcn_f32.err
for f in driver default dcn dcn_f32; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    s=j['step_us']
    print('$f', round(j['value']/1e6,2),'M rows/s', 'wall', round(s['wall_us']), 'gpu', round(s['gpu_us']), 'median', round(s['median'],1), 'parity', j.get('parity',{}).get('ok'), j.get('config',{}).get('tower_mfma'))
    print('   kernels', {k: round(v,1) for k,v in (j.get('kernels_us') or {}).items()})
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
