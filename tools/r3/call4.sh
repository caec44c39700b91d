#!/bin/bash
# round 3, GPU call 4: epilogue as per-wave work pulling, the matrix waves join after their GEMM
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c4
timeout 1200 python -m pytest tests/test_fused_gpu.py "tests/test_headline_gpu.py::test_headline_config_matches_oracle" "tests/test_headline_gpu.py::test_headline_rows_in_step_equals_separate_optimizer_step" -x -q -m gpu 2>&1 | tail -25 > ${O}_tests.txt
tail -4 ${O}_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 > ${O}_line_join.json 2> ${O}_line_join.err
DT_ROWS_JOIN=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 > ${O}_line_nojoin.json 2> ${O}_line_nojoin.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 --dist zipf > ${O}_line_zipf.json 2> ${O}_line_zipf.err
for f in join nojoin zipf; do echo $f; cut -c1-200 ${O}_line_$f.json; tail -1 ${O}_line_$f.err; done
timeout 400 bash tools_prof.sh r3c4_prof --steps 100 --warmup 10 --no-parity > ${O}_stats.txt 2>&1
head -8 ${O}_stats.txt
ROWS=1 timeout 300 python tools/phase_times.py > ${O}_stamps_rows.txt 2>&1
grep -A 3 "k_mlp_bwd stamps" ${O}_stamps_rows.txt; grep -A 8 "k_wgrad stamps" ${O}_stamps_rows.txt
