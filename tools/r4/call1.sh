#!/bin/bash
# round 4, call 1: the new tests (compiled fit, DCN under DP, in-step vs oracle), the whole GPU suite, bench lines through the
# product's compiled loop (driver command + default), DCN --force-dp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c1
O=gpurun_out/r4c1
timeout 300 python -m pytest tests/test_compiled_gpu.py -x -q 2>&1 | tail -25 > $O/t_compiled.txt
timeout 400 python -m pytest tests/test_headline_gpu.py tests/test_parallel_gpu.py -q -k "two_steps or two_process" 2>&1 | tail -25 > $O/t_new.txt
timeout 300 python -m pytest tests/test_fused_gpu.py -q -k "rccl or three_steps" 2>&1 | tail -25 > $O/t_fused_new.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
python bench.py --no-cpu-baseline > $O/line_default.json 2> $O/line_default.err
python bench.py --model DCN --force-dp --no-cpu-baseline --no-parity > $O/line_dcn_dp.json 2> $O/line_dcn_dp.err
python bench.py --model DCN --force-dp --dist zipf --no-cpu-baseline --no-parity > $O/line_dcn_dp_zipf.json 2> $O/line_dcn_dp_zipf.err
python bench.py --model DCN --no-cpu-baseline --no-parity > $O/line_dcn.json 2> $O/line_dcn.err
timeout 400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/t_all.txt
tail -5 $O/t_compiled.txt $O/t_new.txt $O/t_fused_new.txt $O/t_all.txt
for f in driver default dcn_dp dcn_dp_zipf dcn; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    print('$f', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', j['step_us'], j.get('first_replay_us'), j['config'].get('steps_per_graph_replay'), j.get('fit_rows_per_s'), j.get('fwd_bwd_only_rows_per_s'), j.get('phases'), (j.get('parity') or {}).get('ok'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
