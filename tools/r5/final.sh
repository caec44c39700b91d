#!/bin/bash
# last call of round 5: bench.py now creates its timing events before the timed region (call 18) and takes 20 steps per
# replay by default (call 21): the compiled-loop tests, then refresh.sh (PMC passes, the two DeepFM lines, the kernel table)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_compiled_gpu.py -m gpu -x -q > gpurun_out/r05_final_tests.txt 2>&1
tail -2 gpurun_out/r05_final_tests.txt
bash tools/r5/refresh.sh
python bench.py --dist zipf --no-cpu-baseline > gpurun_out/r05_line_zipf.json 2> gpurun_out/r05_line_zipf.err
DT_AMD_CHAIN=0 python bench.py --no-cpu-baseline --no-parity > gpurun_out/r05_line_deepfm_nochain.json 2> gpurun_out/r05_line_deepfm_nochain.err
python bench.py --model DCN --no-cpu-baseline > gpurun_out/r05_line_dcn.json 2> gpurun_out/r05_line_dcn.err
for f in zipf deepfm_nochain dcn; do grep "^{" gpurun_out/r05_line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'parity', (j.get('parity') or {}).get('ok'))"; done
