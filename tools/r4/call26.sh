#!/bin/bash
# round 4, call 26: the PMC passes + the two DeepFM lines on the final source hash (after call 25's loss block)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools_pmc.sh r04_pmc_fetch FETCH_SIZE --steps 20 --warmup 5 --no-parity > gpurun_out/r04_pmc_fetch.txt 2>&1
bash tools_pmc.sh r04_pmc_write WRITE_SIZE --steps 20 --warmup 5 --no-parity > gpurun_out/r04_pmc_write.txt 2>&1
python tools/make_traffic.py gpurun_out/r04_pmc_fetch/r04_pmc_fetch_counter_collection.csv gpurun_out/r04_pmc_write/r04_pmc_write_counter_collection.csv gpurun_out/r04_pmc_fetch.log profiles/r03_counter_calibration.json > gpurun_out/r04_traffic_stdout.txt 2>&1
cp profiles/deepfm_traffic.json gpurun_out/deepfm_traffic.json
tail -2 gpurun_out/r04_traffic_stdout.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_line_driver.json 2> gpurun_out/r04_line_driver.err
python bench.py > gpurun_out/r04_line_deepfm.json 2> gpurun_out/r04_line_deepfm.err
for f in driver deepfm; do grep "^{" gpurun_out/r04_line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'frac', round(j['roofline']['frac'],4), 'traffic', j['roofline'].get('traffic'), 'parity', (j.get('parity') or {}).get('ok'), 'fit', j.get('fit_rows_per_s'))" || tail -3 gpurun_out/r04_line_$f.err; done
