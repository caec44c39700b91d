#!/bin/bash
# round 4, call 15: kernel stats of the DeepFM / DCN steps after the split-bf16 DCN tile kernel and the self-advancing gather
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c15
O=gpurun_out/r4c15
export TMPDIR=/tmp
bash tools_prof.sh r4c15_deepfm --steps 100 --warmup 10 --no-parity > $O/stats_deepfm.txt 2>&1
bash tools_prof.sh r4c15_dcn --model DCN --steps 100 --warmup 10 --no-parity > $O/stats_dcn.txt 2>&1
bash tools_prof.sh r4c15_dcn_f32 --model DCN --tower f32 --steps 100 --warmup 10 --no-parity > $O/stats_dcn_f32.txt 2>&1
head -14 $O/stats_deepfm.txt; head -14 $O/stats_dcn.txt; head -14 $O/stats_dcn_f32.txt
