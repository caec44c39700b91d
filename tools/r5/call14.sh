#!/bin/bash
# round 5, call 14: two segments per iteration in the segment walk — optimizer / fused / headline suites, Zipf and uniform lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_compiled_gpu.py -m gpu -x -q > gpurun_out/c14_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/c14_tests.txt | cut -c1-300
bash tools_prof.sh c14_zipf --dist zipf --steps 100 --warmup 10 --no-parity | head -5
bash tools_prof.sh c14_uniform --steps 100 --warmup 10 --no-parity | head -5
bash tools_prof.sh c14_b65536 --batch 65536 --steps 30 --warmup 10 --no-parity | head -6
bash tools_prof.sh c14_autoint --model AutoInt --steps 50 --warmup 10 --no-parity | head -4
line() { grep "^{" $1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$2', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), j.get('phases'))" || tail -5 ${1%.json}.err; }
timeout 300 python bench.py --dist zipf --no-cpu-baseline --no-extras --no-parity > gpurun_out/c14_zipf.json 2> gpurun_out/c14_zipf.err; line gpurun_out/c14_zipf.json zipf
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-parity > gpurun_out/c14_uniform.json 2> gpurun_out/c14_uniform.err; line gpurun_out/c14_uniform.json uniform
timeout 300 python bench.py --force-sharded --no-cpu-baseline --no-extras --no-parity > gpurun_out/c14_sharded.json 2> gpurun_out/c14_sharded.err; line gpurun_out/c14_sharded.json sharded_w1
