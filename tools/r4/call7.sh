#!/bin/bash
# round 4, call 7: pre-election inside the compiled loop; one-pass x3 layouts in k_prep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c7
O=gpurun_out/r4c7
timeout 300 python -m pytest tests/test_compiled_gpu.py tests/test_x3_gpu.py -q -x 2>&1 | tail -30 > $O/t_a.txt
python bench.py --no-cpu-baseline > $O/line_default.json 2> $O/line_default.err
DT_AMD_PREELECT=0 python bench.py --no-cpu-baseline --no-parity > $O/line_nopre.json 2> $O/line_nopre.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
python bench.py --no-cpu-baseline --dist zipf > $O/line_zipf.json 2> $O/line_zipf.err
python bench.py --no-cpu-baseline --model DCN > $O/line_dcn.json 2> $O/line_dcn.err
bash tools_prof.sh r4c7_deepfm --steps 100 --warmup 10 --no-parity > $O/stats.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/t_all.txt
tail -n 5 $O/t_a.txt; grep -n "passed\|failed\|FAILED" $O/t_all.txt | head; head -9 $O/stats.txt
for f in default nopre driver zipf dcn; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', round(j['step_us']['median'],1), j.get('first_replay_us'), j.get('fit_rows_per_s'), j.get('fwd_bwd_only_rows_per_s'), p.get('ok'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
