#!/bin/bash
# round 3, GPU call 3: R as its own 16-wave kernel, permlane sums in the dXn GEMM, E' + dense Adam + segments + advance in one launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c3
timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_optim_gpu.py "tests/test_headline_gpu.py::test_headline_config_matches_oracle" "tests/test_headline_gpu.py::test_headline_rows_in_step_equals_separate_optimizer_step" -x -q -m gpu 2>&1 | tail -25 > ${O}_tests.txt
tail -8 ${O}_tests.txt
timeout 300 python bench.py --no-cpu-baseline --steps 200 > ${O}_line_pipe.json 2> ${O}_line_pipe.err
DT_AMD_STEP_IN_STEP=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 > ${O}_line_rowsonly.json 2> ${O}_line_rowsonly.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 --dist zipf > ${O}_line_pipe_zipf.json 2> ${O}_line_pipe_zipf.err
DT_ADAM_SEG_BLOCKS=2048 timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 --dist zipf > ${O}_line_pipe_zipf2k.json 2> ${O}_line_pipe_zipf2k.err
for f in pipe rowsonly pipe_zipf pipe_zipf2k; do echo $f; cut -c1-260 ${O}_line_$f.json; tail -2 ${O}_line_$f.err; done
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c3_line_pipe.json').read().strip().splitlines()[-1])
print('parity ok', j.get('parity',{}).get('ok')); print('step_us', j['step_us'], 'fwd_bwd_only', j.get('fwd_bwd_only_rows_per_s'))
PY
timeout 400 bash tools_prof.sh r3c3_prof_pipe --steps 100 --warmup 10 --no-parity > ${O}_stats_pipe.txt 2>&1
timeout 400 bash tools_prof.sh r3c3_prof_zipf --steps 100 --warmup 10 --no-parity --dist zipf > ${O}_stats_zipf.txt 2>&1
head -9 ${O}_stats_pipe.txt; head -9 ${O}_stats_zipf.txt
ROWS=1 timeout 300 python tools/phase_times.py > ${O}_stamps_rows.txt 2>&1
grep -A 14 "k_mlp_fwd stamps" ${O}_stamps_rows.txt | tail -6; grep -A 3 "k_mlp_bwd stamps" ${O}_stamps_rows.txt; grep -A 7 "k_wgrad stamps" ${O}_stamps_rows.txt
