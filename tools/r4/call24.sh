#!/bin/bash
# round 4, call 24 (after the evidence pass): the CIN kernels as separate four-wave / eight-wave kernels again; CIN parity, the
# xDeepFM lines and stats, and the PMC passes + the two DeepFM lines on the final source hash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r04b
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_x3_gpu.py tests/test_bf16_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "cin or xdeepfm or bf16" > ${O}_pytest.log 2>&1
grep -E "passed|failed|error|FAILED" ${O}_pytest.log | tail -4
bash tools_pmc.sh r04_pmc_fetch FETCH_SIZE --steps 20 --warmup 5 --no-parity > gpurun_out/r04_pmc_fetch.txt 2>&1
bash tools_pmc.sh r04_pmc_write WRITE_SIZE --steps 20 --warmup 5 --no-parity > gpurun_out/r04_pmc_write.txt 2>&1
python tools/make_traffic.py gpurun_out/r04_pmc_fetch/r04_pmc_fetch_counter_collection.csv gpurun_out/r04_pmc_write/r04_pmc_write_counter_collection.csv gpurun_out/r04_pmc_fetch.log profiles/r03_counter_calibration.json > gpurun_out/r04_traffic_stdout.txt 2>&1
cp profiles/deepfm_traffic.json gpurun_out/deepfm_traffic.json
tail -3 gpurun_out/r04_traffic_stdout.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_line_driver.json 2> gpurun_out/r04_line_driver.err
python bench.py > gpurun_out/r04_line_deepfm.json 2> gpurun_out/r04_line_deepfm.err
python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r04_line_xdeepfm.json 2> gpurun_out/r04_line_xdeepfm.err
python bench.py --model xDeepFM --cin bf16 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r04_line_xdeepfm_bf16.json 2> gpurun_out/r04_line_xdeepfm_bf16.err
bash tools_prof.sh r04_xdeepfm_x3 --model xDeepFM --steps 20 --warmup 5 --no-parity > gpurun_out/r04_stats_xdeepfm_x3.txt 2>&1
DT_CIN_WIDE=0 DT_CIN_WGRAD_WIDE=0 bash tools_prof.sh r04_xdeepfm_x3_narrow --model xDeepFM --steps 20 --warmup 5 --no-parity > gpurun_out/r04_stats_xdeepfm_x3_narrow.txt 2>&1
for f in driver deepfm xdeepfm xdeepfm_bf16; do grep "^{" gpurun_out/r04_line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'frac', round(j['roofline']['frac'],4), 'traffic', j['roofline'].get('traffic'), 'parity', (j.get('parity') or {}).get('ok'))" || tail -3 gpurun_out/r04_line_$f.err; done
head -6 gpurun_out/r04_stats_xdeepfm_x3.txt; head -5 gpurun_out/r04_stats_xdeepfm_x3_narrow.txt
