#!/bin/bash
# round 3, GPU call 9: the row update takes p from the X tile (no second random read without dropout)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c9
timeout 1200 python -m pytest tests/test_fused_gpu.py "tests/test_headline_gpu.py::test_headline_config_matches_oracle" "tests/test_headline_gpu.py::test_headline_rows_in_step_equals_separate_optimizer_step" -x -q -m gpu 2>&1 | tail -25 > ${O}_tests.txt
tail -4 ${O}_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 > ${O}_line.json 2> ${O}_line.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --dist zipf > ${O}_line_zipf.json 2> ${O}_line_zipf.err
for f in line line_zipf; do echo $f; cut -c1-200 ${O}_$f.json; tail -1 ${O}_$f.err; done
timeout 400 bash tools_prof.sh r3c9_prof --steps 100 --warmup 10 --no-parity > ${O}_stats.txt 2>&1
head -7 ${O}_stats.txt
