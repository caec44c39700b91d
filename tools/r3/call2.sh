#!/bin/bash
# round 3, GPU call 2: the pipelined DeepFM step (dXn GEMM in C, R, [E | D + row Adam], E', O') — parity + A/B timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c2
timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_optim_gpu.py "tests/test_headline_gpu.py::test_headline_config_matches_oracle" "tests/test_headline_gpu.py::test_headline_rows_in_step_equals_separate_optimizer_step" -x -q -m gpu 2>&1 | tail -25 > ${O}_tests.txt
tail -25 ${O}_tests.txt
timeout 300 python bench.py --no-cpu-baseline --steps 200 > ${O}_line_pipe.json 2> ${O}_line_pipe.err
DT_AMD_ROWS_IN_STEP=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 > ${O}_line_pipe_norows.json 2> ${O}_line_pipe_norows.err
DT_STEP_PIPE=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 > ${O}_line_old.json 2> ${O}_line_old.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 --dist zipf > ${O}_line_pipe_zipf.json 2> ${O}_line_pipe_zipf.err
for f in pipe pipe_norows old pipe_zipf; do echo $f; cut -c1-260 ${O}_line_$f.json; tail -3 ${O}_line_$f.err; done
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c2_line_pipe.json').read().strip().splitlines()[-1])
print('parity', json.dumps(j.get('parity'))[:1800]); print('step_us', j['step_us'], 'fwd_bwd_only', j.get('fwd_bwd_only_rows_per_s'))
PY
timeout 400 bash tools_prof.sh r3c2_prof_pipe --steps 100 --warmup 10 --no-parity > ${O}_stats_pipe.txt 2>&1
DT_AMD_ROWS_IN_STEP=0 timeout 400 bash tools_prof.sh r3c2_prof_norows --steps 100 --warmup 10 --no-parity > ${O}_stats_norows.txt 2>&1
timeout 400 bash tools_prof.sh r3c2_prof_zipf --steps 100 --warmup 10 --no-parity --dist zipf > ${O}_stats_zipf.txt 2>&1
head -12 ${O}_stats_pipe.txt; head -10 ${O}_stats_norows.txt; head -10 ${O}_stats_zipf.txt
ROWS=1 timeout 300 python tools/phase_times.py > ${O}_stamps_rows.txt 2>&1
tail -40 ${O}_stamps_rows.txt
