#!/bin/bash
# round 6, call 3: N > 1 step structures at world size 1 through RCCL with the WHOLE step (exchange + optimizer) captured
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in dp sharded; do
  timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras --force-$v > $O/$v.json 2> $O/$v.err
  DT_AMD_DP_GRAPH=0 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras --force-$v > $O/${v}_split.json 2> $O/${v}_split.err
done
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras --force-dp --model DCN > $O/dp_dcn.json 2> $O/dp_dcn.err
timeout 1200 python -m pytest tests/test_parallel_gpu.py tests/test_compiled_gpu.py -x -q -m gpu > $O/pytest_par.txt 2>&1
tail -4 $O/pytest_par.txt
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "rank or parallel or sharded" > $O/pytest_rank.txt 2>&1
tail -4 $O/pytest_rank.txt
timeout 1200 python -m pytest tests/test_reference_models_gpu.py tests/test_models_gpu.py -x -q -m gpu > $O/pytest_models.txt 2>&1
tail -4 $O/pytest_models.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c3/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        su=j['step_us']
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,1), 'M; rep', su.get('repeat_step_us'), 'whole', j['config'].get('dp_whole_step_graph'), 'spg', j['config'].get('steps_per_graph_replay'), 'phases', j.get('phases'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
