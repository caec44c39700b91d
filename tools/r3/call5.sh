#!/bin/bash
# round 3, GPU call 5: dXn GEMM on 32x32x2 tiles, segment walk with prefetched records, lr_t by direct load in k_finish_step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c5
timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_optim_gpu.py "tests/test_headline_gpu.py::test_headline_config_matches_oracle" "tests/test_headline_gpu.py::test_headline_rows_in_step_equals_separate_optimizer_step" -x -q -m gpu 2>&1 | tail -25 > ${O}_tests.txt
tail -4 ${O}_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 > ${O}_line.json 2> ${O}_line.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 --dist zipf > ${O}_line_zipf.json 2> ${O}_line_zipf.err
DT_ADAM_SEG_BLOCKS=512 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 > ${O}_line_s512.json 2> ${O}_line_s512.err
DT_ADAM_SEG_BLOCKS=2048 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --dist zipf > ${O}_line_zipf_s2k.json 2> ${O}_line_zipf_s2k.err
DT_ADAM_SEG_BLOCKS=3328 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --dist zipf > ${O}_line_zipf_s3k.json 2> ${O}_line_zipf_s3k.err
for f in line line_zipf line_s512 line_zipf_s2k line_zipf_s3k; do echo $f; cut -c1-200 ${O}_$f.json; tail -1 ${O}_$f.err; done
timeout 400 bash tools_prof.sh r3c5_prof --steps 100 --warmup 10 --no-parity > ${O}_stats.txt 2>&1
timeout 400 bash tools_prof.sh r3c5_prof_zipf --steps 100 --warmup 10 --no-parity --dist zipf > ${O}_stats_zipf.txt 2>&1
head -8 ${O}_stats.txt; head -8 ${O}_stats_zipf.txt
ROWS=1 timeout 300 python tools/phase_times.py > ${O}_stamps_rows.txt 2>&1
grep -A 14 "k_mlp_fwd stamps" ${O}_stamps_rows.txt | tail -5
