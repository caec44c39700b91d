#!/bin/bash
# round 4, call 2: re-run of the failed tests, the whole GPU suite, driver line, kernel stats of the default run (the gather's cost)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c2
O=gpurun_out/r4c2
timeout 300 python -m pytest tests/test_compiled_gpu.py -q 2>&1 | tail -40 > $O/t_compiled.txt
timeout 400 python -m pytest tests/test_headline_gpu.py -q -k "two_steps" 2>&1 | tail -25 > $O/t_new.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver.json 2> $O/line_driver.err
timeout 500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/t_all.txt
bash tools_prof.sh r4c2_deepfm --steps 100 --warmup 10 --no-parity > $O/stats_deepfm.txt 2>&1
tail -n 6 $O/t_compiled.txt; tail -n 6 $O/t_new.txt; tail -n 8 $O/t_all.txt; cat $O/stats_deepfm.txt | head -24
python - <<PY
import json
j=json.loads([l for l in open('$O/line_driver.json') if l.startswith('{')][-1])
print('driver', round(j['value']/1e6,2),'M rows/s', round(j['ms_per_step']*1e3,1),'us', j['step_us'], j.get('first_replay_us'), j.get('fit_rows_per_s'), j.get('fit_note'), j.get('extras_error'), j.get('fwd_bwd_only_rows_per_s'))
PY
