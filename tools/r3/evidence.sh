#!/bin/bash
# round 3 evidence in ONE gpurun call: counter calibration -> PMC passes -> traffic json -> bench lines -> kernel stats -> stamps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r03
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_cal_fetch -o cal -- $GRAFT_REPO_ROOT/tools/ub_gather > $GRAFT_REPO_ROOT/${O}_cal_fetch.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_cal_write -o cal -- $GRAFT_REPO_ROOT/tools/ub_gather > $GRAFT_REPO_ROOT/${O}_cal_write.log 2>&1 )
python tools/make_calibration.py $(find gpurun_out/r03_cal_fetch -name '*counter_collection.csv' | head -1) $(find gpurun_out/r03_cal_write -name '*counter_collection.csv' | head -1) > ${O}_counter_calibration.json 2> ${O}_cal.err
cat ${O}_counter_calibration.json | head -60; tail -3 ${O}_cal.err
bash tools_pmc.sh r03_pmc_fetch FETCH_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_fetch.txt 2>&1
bash tools_pmc.sh r03_pmc_write WRITE_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_write.txt 2>&1
python tools/make_traffic.py gpurun_out/r03_pmc_fetch/r03_pmc_fetch_counter_collection.csv gpurun_out/r03_pmc_write/r03_pmc_write_counter_collection.csv gpurun_out/r03_pmc_fetch.log ${O}_counter_calibration.json > ${O}_traffic_stdout.txt 2>&1
cp profiles/deepfm_traffic.json gpurun_out/deepfm_traffic.json
tail -5 ${O}_traffic_stdout.txt
python bench.py > ${O}_line_deepfm.json 2> ${O}_line_deepfm.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${O}_line_deepfm_s20.json 2> ${O}_line_deepfm_s20.err
python bench.py --dist zipf --no-cpu-baseline > ${O}_line_zipf.json 2> ${O}_line_zipf.err
python bench.py --model DCN --no-cpu-baseline > ${O}_line_dcn.json 2> ${O}_line_dcn.err
python bench.py --steps-per-graph 1 --no-cpu-baseline --no-parity > ${O}_line_deepfm_spg1.json 2> ${O}_line_deepfm_spg1.err
bash tools_prof.sh r03_deepfm --steps 100 --warmup 10 --no-parity > ${O}_stats_deepfm.txt 2>&1
bash tools_prof.sh r03_deepfm_zipf --steps 100 --warmup 10 --no-parity --dist zipf > ${O}_stats_deepfm_zipf.txt 2>&1
bash tools_prof.sh r03_dcn --model DCN --steps 100 --warmup 10 --no-parity > ${O}_stats_dcn.txt 2>&1
ROWS=1 timeout 200 python tools/phase_times.py > ${O}_deepfm_phase_stamps.txt 2>&1
MODEL=DCN ROWS=1 timeout 200 python tools/phase_times.py > ${O}_dcn_phase_stamps.txt 2>&1
for f in deepfm deepfm_s20 zipf dcn deepfm_spg1; do grep "^{" ${O}_line_$f.json | cut -c1-230; done
head -8 ${O}_stats_deepfm.txt
