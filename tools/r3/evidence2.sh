#!/bin/bash
# round 3 final evidence in ONE gpurun call (supersedes evidence.sh's outputs): counter calibration -> PMC passes -> traffic
# json -> bench lines (all four headline models, both id distributions, the N>1 step structures at world size 1) -> kernel
# stats -> stamps -> microbenchmark
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r03
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_cal_fetch -o cal -- $GRAFT_REPO_ROOT/tools/ub_gather > $GRAFT_REPO_ROOT/${O}_cal_fetch.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_cal_write -o cal -- $GRAFT_REPO_ROOT/tools/ub_gather > $GRAFT_REPO_ROOT/${O}_cal_write.log 2>&1 )
python tools/make_calibration.py $(find gpurun_out/r03_cal_fetch -name '*counter_collection.csv' | head -1) $(find gpurun_out/r03_cal_write -name '*counter_collection.csv' | head -1) > ${O}_counter_calibration.json 2> ${O}_cal.err
tail -3 ${O}_cal.err
bash tools_pmc.sh r03_pmc_fetch FETCH_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_fetch.txt 2>&1
bash tools_pmc.sh r03_pmc_write WRITE_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_write.txt 2>&1
python tools/make_traffic.py gpurun_out/r03_pmc_fetch/r03_pmc_fetch_counter_collection.csv gpurun_out/r03_pmc_write/r03_pmc_write_counter_collection.csv gpurun_out/r03_pmc_fetch.log ${O}_counter_calibration.json > ${O}_traffic_stdout.txt 2>&1
cp profiles/deepfm_traffic.json gpurun_out/deepfm_traffic.json
tail -3 ${O}_traffic_stdout.txt
python bench.py > ${O}_line_deepfm.json 2> ${O}_line_deepfm.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${O}_line_deepfm_s20.json 2> ${O}_line_deepfm_s20.err
python bench.py --dist zipf --no-cpu-baseline > ${O}_line_zipf.json 2> ${O}_line_zipf.err
python bench.py --model DCN --no-cpu-baseline > ${O}_line_dcn.json 2> ${O}_line_dcn.err
python bench.py --steps-per-graph 1 --no-cpu-baseline --no-parity > ${O}_line_deepfm_spg1.json 2> ${O}_line_deepfm_spg1.err
python bench.py --steps-per-graph 10 --no-cpu-baseline --no-parity > ${O}_line_deepfm_spg10.json 2> ${O}_line_deepfm_spg10.err
python bench.py --model xDeepFM --steps 20 --warmup 3 --no-cpu-baseline > ${O}_line_xdeepfm.json 2> ${O}_line_xdeepfm.err
DT_AMD_CIN_DTYPE=bf16 python bench.py --model xDeepFM --steps 20 --warmup 3 --no-cpu-baseline > ${O}_line_xdeepfm_bf16.json 2> ${O}_line_xdeepfm_bf16.err
python bench.py --model AutoInt --steps 50 --warmup 5 --no-cpu-baseline > ${O}_line_autoint.json 2> ${O}_line_autoint.err
python bench.py --force-dp --no-cpu-baseline --no-parity > ${O}_line_dp_w1.json 2> ${O}_line_dp_w1.err
python bench.py --force-dp --dist zipf --no-cpu-baseline --no-parity > ${O}_line_dp_w1_zipf.json 2> ${O}_line_dp_w1_zipf.err
python bench.py --force-sharded --no-cpu-baseline --no-parity > ${O}_line_sharded_w1.json 2> ${O}_line_sharded_w1.err
bash tools_prof.sh r03_deepfm --steps 100 --warmup 10 --no-parity > ${O}_stats_deepfm.txt 2>&1
bash tools_prof.sh r03_deepfm_zipf --steps 100 --warmup 10 --no-parity --dist zipf > ${O}_stats_deepfm_zipf.txt 2>&1
bash tools_prof.sh r03_dcn --model DCN --steps 100 --warmup 10 --no-parity > ${O}_stats_dcn.txt 2>&1
bash tools_prof.sh r03_xdeepfm_f32 --model xDeepFM --steps 20 --warmup 3 --no-parity > ${O}_stats_xdeepfm_f32.txt 2>&1
DT_AMD_CIN_DTYPE=bf16 bash tools_prof.sh r03_xdeepfm_bf16 --model xDeepFM --steps 20 --warmup 3 --no-parity > ${O}_stats_xdeepfm_bf16.txt 2>&1
bash tools_prof.sh r03_autoint --model AutoInt --steps 50 --warmup 5 --no-parity > ${O}_stats_autoint.txt 2>&1
ROWS=1 timeout 200 python tools/phase_times.py > ${O}_deepfm_phase_stamps.txt 2>&1
MODEL=DCN ROWS=1 timeout 200 python tools/phase_times.py > ${O}_dcn_phase_stamps.txt 2>&1
for f in deepfm deepfm_s20 zipf dcn deepfm_spg1 deepfm_spg10 xdeepfm xdeepfm_bf16 autoint dp_w1 dp_w1_zipf sharded_w1; do grep "^{" ${O}_line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'frac', round(j['roofline']['frac'],4), 'parity', (j.get('parity') or {}).get('ok'), j.get('phases'))" || tail -3 ${O}_line_$f.err; done
head -8 ${O}_stats_deepfm.txt
