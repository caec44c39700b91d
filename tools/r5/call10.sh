#!/bin/bash
# round 5, call 10: A/B build — GEMM1 operand queue 8 deep (kPF), finishing launch without the 4-waves-per-SIMD register bound
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools_prof.sh c10_deepfm --steps 100 --warmup 10 --no-parity | head -6
bash tools_prof.sh c10_dcn --model DCN --steps 100 --warmup 10 --no-parity | head -6
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-parity > gpurun_out/c10_deepfm.json 2> gpurun_out/c10_deepfm.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity > gpurun_out/c10_driver.json 2> gpurun_out/c10_driver.err
for f in deepfm driver; do grep "^{" gpurun_out/c10_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), j['config']['steps_per_graph_replay'])"; done
