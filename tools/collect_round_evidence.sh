#!/bin/bash
# The round's evidence in ONE gpurun call (round 2 ran exactly this: 120 s on the box):
#   PMC passes -> profiles/deepfm_traffic.json -> bench lines (the default one last-but-first so it carries the traffic) ->
#   kernel stats -> DCN phase stamps -> the GPU test suite.  Outputs land in gpurun_out/ and are copied to profiles/ by hand.
#   usage: gpurun --timeout 330 -- 'bash tools/collect_round_evidence.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools_pmc.sh r02_pmc_fetch FETCH_SIZE --steps 20 --warmup 3 --no-parity > gpurun_out/final_pmc_fetch.txt 2>&1
bash tools_pmc.sh r02_pmc_write WRITE_SIZE --steps 20 --warmup 3 --no-parity > gpurun_out/final_pmc_write.txt 2>&1
python tools/make_traffic.py gpurun_out/r02_pmc_fetch/r02_pmc_fetch_counter_collection.csv gpurun_out/r02_pmc_write/r02_pmc_write_counter_collection.csv gpurun_out/r02_pmc_fetch.log > gpurun_out/final_traffic_stdout.txt 2>&1
cp profiles/deepfm_traffic.json gpurun_out/deepfm_traffic.json
python bench.py > gpurun_out/final_line_deepfm.json 2> gpurun_out/final_line_deepfm.err
python bench.py --dist zipf --no-cpu-baseline > gpurun_out/final_line_zipf.json 2> gpurun_out/final_line_zipf.err
python bench.py --model DCN --no-cpu-baseline > gpurun_out/final_line_dcn.json 2> gpurun_out/final_line_dcn.err
bash tools_prof.sh r02_deepfm --steps 100 --warmup 10 --no-parity > gpurun_out/final_stats_deepfm.txt 2>&1
bash tools_prof.sh r02_dcn --model DCN --steps 50 --warmup 8 --no-parity > gpurun_out/final_stats_dcn.txt 2>&1
MODEL=DCN DT_AMD_STEP_STAMPS=1 timeout 100 python tools/phase_times.py > gpurun_out/final_dcn_stamps.txt 2>&1
timeout 240 python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/final_tests.txt
tail -3 gpurun_out/final_tests.txt; cut -c1-200 gpurun_out/final_line_deepfm.json; grep -o '"traffic_over_algorithmic": [0-9.]*' gpurun_out/deepfm_traffic.json
