#!/bin/bash
# round 3, GPU call 7: finishing launch with prefetched dense slots, 5 train steps per captured graph by default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c7
timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_optim_gpu.py "tests/test_headline_gpu.py::test_headline_config_matches_oracle" "tests/test_headline_gpu.py::test_headline_rows_in_step_equals_separate_optimizer_step" -x -q -m gpu 2>&1 | tail -25 > ${O}_tests.txt
tail -4 ${O}_tests.txt
timeout 300 python bench.py --no-cpu-baseline --steps 200 > ${O}_line.json 2> ${O}_line.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 20 --warmup 7 > ${O}_line_s20.json 2> ${O}_line_s20.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --dist zipf > ${O}_line_zipf.json 2> ${O}_line_zipf.err
for f in line line_s20 line_zipf; do echo $f; cut -c1-200 ${O}_$f.json; tail -1 ${O}_$f.err; done
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c7_line.json').read().strip().splitlines()[-1])
print('parity ok', j.get('parity',{}).get('ok')); print('step_us', j['step_us'], 'fwd_bwd_only', j.get('fwd_bwd_only_rows_per_s'), j['roofline']['frac'])
PY
timeout 400 bash tools_prof.sh r3c7_prof --steps 100 --warmup 10 --no-parity > ${O}_stats.txt 2>&1
head -8 ${O}_stats.txt
ROWS=1 timeout 300 python tools/phase_times.py > ${O}_stamps_rows.txt 2>&1
grep -A 14 "k_mlp_fwd stamps" ${O}_stamps_rows.txt | tail -5
