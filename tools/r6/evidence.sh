#!/bin/bash
# round 6 evidence in ONE gpurun call: PMC passes -> traffic json -> bench lines (driver command first) -> kernel stats ->
# phase stamps -> MFMA-busy counters -> the GPU test suite.  Outputs: gpurun_out/r06_*; tools/r6/collect_profiles.py copies the
# summaries to profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06
export TMPDIR=/tmp
bash tools_pmc.sh r06_pmc_fetch FETCH_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_fetch.txt 2>&1
bash tools_pmc.sh r06_pmc_write WRITE_SIZE --steps 20 --warmup 5 --no-parity > ${O}_pmc_write.txt 2>&1
python tools/make_traffic.py gpurun_out/r06_pmc_fetch/r06_pmc_fetch_counter_collection.csv gpurun_out/r06_pmc_write/r06_pmc_write_counter_collection.csv gpurun_out/r06_pmc_fetch.log profiles/r03_counter_calibration.json > ${O}_traffic_stdout.txt 2>&1
cp profiles/deepfm_traffic.json gpurun_out/deepfm_traffic.json
tail -3 ${O}_traffic_stdout.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_line_driver.json 2> ${O}_line_driver.err
python bench.py > ${O}_line_deepfm.json 2> ${O}_line_deepfm.err
python bench.py --dist zipf --no-cpu-baseline > ${O}_line_zipf.json 2> ${O}_line_zipf.err
DT_AMD_CHAIN=0 python bench.py --no-cpu-baseline --no-parity > ${O}_line_deepfm_nochain.json 2> ${O}_line_deepfm_nochain.err
python bench.py --batch 32768 --steps 50 --warmup 10 --no-cpu-baseline > ${O}_line_b32768.json 2> ${O}_line_b32768.err
python bench.py --batch 65536 --steps 30 --warmup 10 --no-cpu-baseline > ${O}_line_b65536.json 2> ${O}_line_b65536.err
python bench.py --tower f32 --no-cpu-baseline --no-parity > ${O}_line_deepfm_f32tower.json 2> ${O}_line_deepfm_f32tower.err
python bench.py --tower bf16 --no-cpu-baseline > ${O}_line_deepfm_bf16tower.json 2> ${O}_line_deepfm_bf16tower.err
python bench.py --model DCN --no-cpu-baseline > ${O}_line_dcn.json 2> ${O}_line_dcn.err
python bench.py --model DCN --force-dp --no-cpu-baseline --no-parity > ${O}_line_dcn_dp_w1.json 2> ${O}_line_dcn_dp_w1.err
python bench.py --force-dp --no-cpu-baseline --no-parity > ${O}_line_dp_w1.json 2> ${O}_line_dp_w1.err
DT_BENCH_BOTH_LAYOUTS=1 python bench.py --force-sharded --no-cpu-baseline --no-parity > ${O}_line_sharded_w1.json 2> ${O}_line_sharded_w1.err
python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline > ${O}_line_xdeepfm.json 2> ${O}_line_xdeepfm.err
python bench.py --model AutoInt --steps 50 --warmup 10 --no-cpu-baseline > ${O}_line_autoint.json 2> ${O}_line_autoint.err
python bench.py --model AutoInt --attn f32 --steps 50 --warmup 10 --no-cpu-baseline --no-parity > ${O}_line_autoint_f32.json 2> ${O}_line_autoint_f32.err
python bench.py --model AutoInt --attn bf16 --steps 50 --warmup 10 --no-cpu-baseline > ${O}_line_autoint_bf16.json 2> ${O}_line_autoint_bf16.err
DT_AMD_DP_GRAPH=0 python bench.py --force-dp --no-cpu-baseline --no-parity > ${O}_line_dp_w1_split.json 2> ${O}_line_dp_w1_split.err
bash tools_prof.sh r06_deepfm --steps 100 --warmup 10 --no-parity > ${O}_stats_deepfm.txt 2>&1
bash tools_prof.sh r06_deepfm_zipf --steps 100 --warmup 10 --no-parity --dist zipf > ${O}_stats_deepfm_zipf.txt 2>&1
bash tools_prof.sh r06_deepfm_b32768 --batch 32768 --steps 50 --warmup 10 --no-parity > ${O}_stats_deepfm_b32768.txt 2>&1
bash tools_prof.sh r06_deepfm_b65536 --batch 65536 --steps 30 --warmup 10 --no-parity > ${O}_stats_deepfm_b65536.txt 2>&1
bash tools_prof.sh r06_dcn --model DCN --steps 100 --warmup 10 --no-parity > ${O}_stats_dcn.txt 2>&1
bash tools_prof.sh r06_xdeepfm --model xDeepFM --steps 20 --warmup 5 --no-parity > ${O}_stats_xdeepfm.txt 2>&1
bash tools_prof.sh r06_autoint --model AutoInt --steps 50 --warmup 10 --no-parity > ${O}_stats_autoint.txt 2>&1
ROWS=1 timeout 200 python tools/phase_times.py > ${O}_deepfm_phase_stamps.txt 2>&1
bash tools_pmc.sh r06_pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 20 --warmup 5 --no-parity > ${O}_pmc_mfma.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > ${O}_tests.txt
python __graft_entry__.py --smoke > ${O}_smoke.txt 2>&1; tail -1 ${O}_smoke.txt
for f in driver deepfm zipf deepfm_nochain b32768 b65536 deepfm_f32tower deepfm_bf16tower dcn dcn_dp_w1 dp_w1 dp_w1_split sharded_w1 xdeepfm autoint autoint_f32 autoint_bf16; do grep "^{" ${O}_line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'frac', round(j['roofline']['frac'],4), 'traffic', j['roofline'].get('traffic'), 'parity', (j.get('parity') or {}).get('ok'), j.get('phases'), j.get('fit_rows_per_s'), (j.get('other_layout') or {}).get('rows_per_s'))" || tail -3 ${O}_line_$f.err; done
head -8 ${O}_stats_deepfm.txt; tail -3 ${O}_tests.txt; grep -A3 "k_tower_x3\|k_wgrad_rows" ${O}_pmc_mfma.txt | head -12
