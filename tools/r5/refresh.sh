#!/bin/bash
# round 5, last call: the PMC passes, the two DeepFM lines and the DeepFM kernel table on the FINAL source hash (after
# tools/r5/evidence.sh only comments changed in csrc/)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05
export TMPDIR=/tmp
bash tools_pmc.sh r05_pmc_fetch FETCH_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_fetch.txt 2>&1
bash tools_pmc.sh r05_pmc_write WRITE_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_write.txt 2>&1
python tools/make_traffic.py gpurun_out/r05_pmc_fetch/r05_pmc_fetch_counter_collection.csv gpurun_out/r05_pmc_write/r05_pmc_write_counter_collection.csv gpurun_out/r05_pmc_fetch.log profiles/r03_counter_calibration.json > ${O}_traffic_stdout.txt 2>&1
cp profiles/deepfm_traffic.json gpurun_out/deepfm_traffic.json
tail -3 ${O}_traffic_stdout.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_line_driver.json 2> ${O}_line_driver.err
python bench.py > ${O}_line_deepfm.json 2> ${O}_line_deepfm.err
bash tools_prof.sh r05_deepfm --steps 100 --warmup 10 --no-parity > ${O}_stats_deepfm.txt 2>&1
for f in driver deepfm; do grep "^{" ${O}_line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'frac', round(j['roofline']['frac'],4), 'traffic', j['roofline'].get('traffic'), 'parity', (j.get('parity') or {}).get('ok'))" || tail -3 ${O}_line_$f.err; done
head -6 ${O}_stats_deepfm.txt
